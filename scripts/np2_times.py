"""Timed runs of composite PPDs (y-stage order experiments): python scripts/np2_times.py 3456 6912:2 6400:2"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
for spec in sys.argv[1:]:
    n, _, kc = spec.partition(":")
    n, kc = int(n), float(kc or 1)
    a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1, k_cutoff=kc), ps, collect=False)
    print(n, "k_cutoff", kc, "R", a["stream_factor"], "sec", round(a["seconds_total"], 2),
          {k: round(v) for k, v in a["kernel_ms"].items()}, "var", repr(a["density_variance"]), flush=True)
