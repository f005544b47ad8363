"""y-stage variants of the field store at PPD=4096 (tuning library): tile width / persistent form; times and bitwise
   equality of the records with the default kernel.
   ZD_LIB_PATH=.../libzeldovich_hip_tuning.so python scripts/yf_variants.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zeldovich_plt_amd.api as zd
WMAP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wmap1new.pow")
N = int(os.environ.get("N", "4096"))
ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
p = zd.make_params(N, icformat="RVZel", profile=1, stream_factor=int(os.environ.get("R", "8")), numblock=64)
nplanes = int(os.environ.get("PLANES", "32"))
store = None
ref = None
VARS = [("default", {}), ("nt ring stores", {"ZD_NT": "32"}), ("nt ring stores + x loads", {"ZD_NT": str(32 | 4)}),
        ("nt record stores", {"ZD_NT": "128"}), ("nt ring st + x ld + rec st", {"ZD_NT": str(32 | 4 | 128)}),
        ("nt E loads", {"ZD_NT": "64"}), ("default", {})]
if os.environ.get("OLDVARS"):
    VARS = [("default", {}), ("same row order", {"ZD_PRUNE": str(7 | 2048)}), ("W/2", {"ZD_YW": str(2 if N == 4096 else 4)}),
            ("persistent", {"ZD_YPERSIST": "1"}), ("default", {}), ("same row order", {"ZD_PRUNE": str(7 | 2048)})]
for name, env in VARS:
    for k in ("ZD_YW", "ZD_YPERSIST", "ZD_PRUNE", "ZD_NT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    plan = zd.Plan(p, ps)
    if store is None:
        store = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
        plan.stage_z(0, store.data_ptr())
        torch.cuda.synchronize()
        plan.stats()
    out = torch.zeros(nplanes * N * N * 32, dtype=torch.uint8, device="cuda")
    plan.stage_x(0, store.data_ptr(), 0, nplanes, out.data_ptr())
    torch.cuda.synchronize()
    plan.stats()
    plan.stage_x(0, store.data_ptr(), 0, nplanes, out.data_ptr())
    torch.cuda.synchronize()
    st = plan.stats()
    ms = st["kernel_ms"]
    scale = (N // 2) / (nplanes // 2)
    if ref is None:
        ref = out.clone()
        same = True
    else:
        CH = 1 << 28
        same = all(bool(torch.equal(ref[i:i + CH], out[i:i + CH])) for i in range(0, out.numel(), CH))
    print("%-28s yfft %.1f ms (%.0f ms/step)   xfft %.1f ms (%.0f ms/step)  records==default: %s" % (
        name, ms["k_yfft"], ms["k_yfft"] * scale, ms["k_xfft"], ms["k_xfft"] * scale, same), flush=True)
    plan.close()
    del out
