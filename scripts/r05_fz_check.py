"""round 5: the fused generator + z FFT of the packed PLT store (zd_kernels_fz.hip, plane-interleaved rows) against the two-kernel
Z stage (ZD_StoreMode = packed) on the same parameters: whole planes of records of both residue passes, every field.
    python scripts/r05_fz_check.py [ppd ...]"""
import ctypes as C
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
fmt = "RVdoubleZel"
dt = zd.RECORD_DTYPES[fmt]
store = None


def planes(n, zs, **kw):
    global store
    p = zd.make_params(n, icformat=fmt, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0, **kw)
    if p.stream_factor <= 0:
        free_b, _ = torch.cuda.mem_get_info()
        p.stream_factor = zd.load_library().zd_choose_stream_factor(C.byref(p), 1, int(free_b) + (0 if store is None else store.numel()) - (24 << 30))
    plan = zd.Plan(p, ps, eig=eig)
    if store is None or store.numel() < plan.exchange_bytes:
        store = None
        torch.cuda.empty_cache()
        store = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
    store.fill_(255)  # NaN bytes: a column nobody wrote shows
    out = torch.empty(n * n * dt.itemsize, dtype=torch.uint8, device="cuda")
    where = {}
    for ps_ in range(plan.passes):
        for lp in range(plan.local_planes):
            where[plan.plane_z(ps_, lp)] = (ps_, lp)
    res, last = {}, None
    for z in sorted(zs, key=lambda z: where[z]):
        pass_, lp = where[z]
        if last != pass_:
            plan.stage_z(pass_, store.data_ptr())
            plan.stage_y(store.data_ptr())
            last = pass_
        plan.stage_x(pass_, store.data_ptr(), lp, 1, out.data_ptr())
        torch.cuda.synchronize()
        res[z] = out.cpu().numpy().view(dt).reshape(n, n).copy()
    st = plan.stats()
    info = dict(R=plan.R, passes=plan.passes, store=plan.store_mode, dv=st["density_variance"], maxd=st["max_disp"])
    plan.close()
    return res, info


for n in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    zs = [0, 1, 2, 3, n // 2 - 1, n // 2, n // 2 + 6, n - 1]
    a, ia = planes(n, zs)
    b, ib = planes(n, zs, store_mode=2)
    print("PPD", n, "fused:", ia, "\n   two kernels:", ib)
    worst = 0.0
    for z in zs:
        assert (a[z]["ijk"] == b[z]["ijk"]).all()
        scale = np.abs(b[z]["d"]).max()
        e = max(np.abs(a[z]["d"] - b[z]["d"]).max(), np.abs(a[z]["v"] - b[z]["v"]).max()) / scale
        print("  plane %5d  max|q| %.4f  max diff / max|q| = %.3e  finite %s" % (z, scale, e, bool(np.isfinite(a[z]["d"]).all())))
        worst = max(worst, e)
    print("PPD", n, "worst", worst, "density_variance rel diff", abs(ia["dv"] - ib["dv"]) / abs(ib["dv"]))
    assert worst < 1e-11, worst
print("OK")
