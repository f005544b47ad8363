"""Round 4 probe: which store combinations a PPD=8192 plan accepts.  Before the fix the plan of four reference arrays was created and the
x stage of the first pass failed (2048 threads per workgroup); since then zd_plan_create refuses it (tests/test_gpu_parity.py
test_ppd8192_on_four_reference_arrays_is_refused_at_plan_creation)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
from oracle import zdo
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
eig = zdo.synthetic_eigenmodes(32)
for kw, e in ((dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, store_mode="reference", stream_factor=256), eig),
              (dict(store_mode="packed", stream_factor=64), None),
              (dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, qdensity=1, stream_factor=256), eig)):
    try:
        pl = zd.Plan(zd.make_params(8192, k_cutoff=2.0, **kw), ps, eig=e)
        print("PLAN OK", kw, "narray", pl.narray, "R", pl.R, "store", pl.store_mode)
        pl.close()
    except Exception as ex:
        print("REFUSED", kw, ex)
# does the x stage of four reference arrays launch at 8192?  (4 lines of 512 threads)
import torch
try:
    pl = zd.Plan(zd.make_params(8192, k_cutoff=2.0, icformat="RVZel", qPLT=1, qPLTrescale=1, PLT_target_z=5.0, store_mode="reference", stream_factor=256), ps, eig=eig)
except RuntimeError as ex:
    raise SystemExit("REFUSED at plan creation: %s" % ex)
store = torch.empty(pl.exchange_bytes, dtype=torch.uint8, device="cuda")
out = torch.empty(8192 * 8192 * 32, dtype=torch.uint8, device="cuda")
try:
    pl.stage_z(0, store.data_ptr()); pl.stage_y(store.data_ptr()); pl.stage_x(0, store.data_ptr(), 0, 1, out.data_ptr()); torch.cuda.synchronize()
    import numpy as np
    rec = out.view(8192, 8192, 32)[::64, ::64].cpu().numpy().view(zd.RECORD_DTYPES["RVZel"])
    print("X STAGE OK", float(np.abs(rec["d"]).max()), pl.stats()["max_disp"])
except Exception as ex:
    print("X STAGE FAILED", ex)
