"""Composite grids: the Z stage's two kernels one after the other (serial_z = 1) and on two streams (0) — does the z FFT run beside the
generator?  python scripts/r05_serial_z_composite.py 3456 4000 5184:2 ..."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
for spec in sys.argv[1:]:
    f = spec.split(":")
    n, kc = int(f[0]), float(f[1]) if len(f) > 1 else 1.0
    for serial in (1, 0):
        a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1, k_cutoff=kc, serial_z=serial), ps, collect=False)
        print(n, "ZA k_cutoff", kc, "R", a["stream_factor"], "serial_z", serial, "sec", round(a["seconds_total"], 2), {k: round(v) for k, v in a["kernel_ms"].items()}, flush=True)
