"""PPD=6912 ZA with the Z stage's two kernels one after the other (serial_z): what each costs alone, at R = 32 (z lines of 216 = 8 * 27)
and R = 36 (192 = 64 * 3)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
for R in (32, 36):
    for serial in (1, 0):
        a = zd.generate(zd.make_params(6912, icformat="RVZel", profile=1, stream_factor=R, serial_z=serial), ps, collect=False)
        print("6912 ZA R", R, "serial_z", serial, "sec", round(a["seconds_total"], 2), {k: round(v) for k, v in a["kernel_ms"].items()}, flush=True)
