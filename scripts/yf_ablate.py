"""Ablation timing of the field-store y stage (k_yfft_f) at PPD=4096: needs the tuning library
   (make -C zeldovich_plt_amd/csrc tuning; ZD_LIB_PATH=.../libzeldovich_hip_tuning.so python scripts/yf_ablate.py)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zeldovich_plt_amd.api as zd
WMAP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wmap1new.pow")
N = int(os.environ.get("N", "4096"))
ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
p = zd.make_params(N, icformat="RVZel", profile=1, stream_factor=int(os.environ.get("R", "8")), numblock=64)
store = None
nplanes = int(os.environ.get("PLANES", "64"))
ALL = [("full", 0), ("no stores", 256), ("no fft", 128), ("one load/row", 512), ("aligned mirror", 1024),
       ("no x array", 4096), ("no fft no stores", 384), ("one load + aligned", 1536), ("full", 0)]
want = os.environ.get("VARIANTS")
for name, bits in [v for v in ALL if not want or v[0] in want.split(",")]:
    os.environ["ZD_PRUNE"] = str(7 | bits)
    plan = zd.Plan(p, ps)
    if store is None:
        store = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
        plan.stage_z(0, store.data_ptr())
        torch.cuda.synchronize()
        plan.stats()
    out = torch.empty(nplanes * N * N * 32, dtype=torch.uint8, device="cuda")
    plan.stage_x(0, store.data_ptr(), 0, nplanes, out.data_ptr())
    torch.cuda.synchronize()
    plan.stats()
    plan.stage_x(0, store.data_ptr(), 0, nplanes, out.data_ptr())
    torch.cuda.synchronize()
    st = plan.stats()
    ms = st["kernel_ms"]
    scale = (N // 2) / (nplanes // 2)   # store planes per step / store planes timed
    print("%-22s yfft %.1f ms (%.0f ms/step)   xfft %.1f ms (%.0f ms/step)" % (name, ms["k_yfft"], ms["k_yfft"] * scale, ms["k_xfft"], ms["k_xfft"] * scale), flush=True)
    plan.close()
    del out
