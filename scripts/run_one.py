import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), 'tests/golden/wmap1new.pow'), 720.0)
n = int(sys.argv[1]); R = int(sys.argv[2])
a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1, stream_factor=R), ps, collect=False)
print("PPD", n, "R", a["stream_factor"], {k: round(v, 1) for k, v in a["kernel_ms"].items()})
