"""PPD=6912 ZA (R = 32): generator workgroups per CU beside the composite z FFT (tuning library, ZD_GEN_WGS) — the 432-thread z-FFT
workgroup of z lines of 216 does not fit beside three 256-thread generator workgroups of 128 registers"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
a = zd.generate(zd.make_params(6912, icformat="RVZel", profile=1, stream_factor=int(os.environ.get("R", "32"))), ps, collect=False)
print("6912 ZA R", a["stream_factor"], "ZD_GEN_WGS", os.environ.get("ZD_GEN_WGS"), "sec", round(a["seconds_total"], 2), {k: round(v) for k, v in a["kernel_ms"].items()}, flush=True)
