"""round 5 debugging aid: the block store after the Z stage, fused kernel (plane-interleaved rows) against the two-kernel stage"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import zeldovich_plt_amd.api as zd  # noqa: E402

zd.load_library()
ps = zd.PowerSpectrum.from_file(bench.WMAP, 720.0)
eig = bench.synthetic_eigenmodes(128)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
res = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pad = 24


def zstage(**kw):
    p = zd.make_params(n, icformat="RVdoubleZel", qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0, **kw)
    plan = zd.Plan(p, ps, eig=eig)
    store = torch.zeros(plan.exchange_bytes // 16, 2, dtype=torch.float64, device="cuda")
    plan.stage_z(res, store.data_ptr())
    torch.cuda.synchronize()
    L = n // plan.R
    plan.close()
    return store, L


a, L = zstage()
b, _ = zstage(store_mode=2)
A = a.view(L // 4, 3, n, n + pad, 4, 2).permute(0, 4, 1, 2, 3, 5).reshape(L, 3, n, n + pad, 2)[..., :n, :]
B = b.view(L, 3, n, n + pad, 2)[..., :n, :]
print("max |B|", float(B.abs().max()), "max |A|", float(A.abs().max()))
for arr in range(3):
    for nm, rows in (("self", slice(0, n // 2)), ("twin", slice(n // 2, n))):
        d = (A[:, arr, rows] - B[:, arr, rows]).abs()
        print("array", arr, nm, "max diff", float(d.max()), "max ref", float(B[:, arr, rows].abs().max()))
# where: per row and per column of array 0, self
d = (A[:, 0, : n // 2] - B[:, 0, : n // 2]).abs().amax(dim=(0, 3))  # [row][x]
print("rows with diff > 1e-9:", int((d.amax(dim=1) > 1e-9).sum()), "of", n // 2, "; cols:", int((d.amax(dim=0) > 1e-9).sum()), "of", n)
print("row 0 (ky = 0, general path) max diff", float(d[0].max()))
r = 5
print("row 5: first columns diff", d[r, :12].cpu().numpy())
print("plane profile at row 5 col 3:", (A[:, 0, r, 3] - B[:, 0, r, 3]).abs().amax(dim=1)[:12].cpu().numpy())
print("A", A[:4, 0, r, 3].cpu().numpy(), "\nB", B[:4, 0, r, 3].cpu().numpy())
# is A a scaled / conjugated / shifted version of B ?
va, vb = A[:, 0, r, 3], B[:, 0, r, 3]
ca, cb = torch.view_as_complex(va.contiguous()), torch.view_as_complex(vb.contiguous())
fa, fb = torch.fft.fft(ca), torch.fft.fft(cb)  # back to the line's k2 (up to conj/ordering conventions)
print("spectrum of the line (|k2| small):\n A", fa[:6].cpu().numpy(), "\n B", fb[:6].cpu().numpy())
print(" ratio A/B at k2 = 1..8:", (fa[1:9] / fb[1:9]).cpu().numpy())
print(" ratio A/B at k2 = L-8..L-1:", (fa[-8:] / fb[-8:]).cpu().numpy())
