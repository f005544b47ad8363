"""The fused generator + z FFT of the packed PLT store (csrc/zd_kernels_fz.hip, round 5: one kernel for src/zeldovich.cpp:333-511 +
src/block_array.cpp:387-414 of the half-space rows, the store's rows holding four planes side by side) against the two-kernel Z
stage on plain rows (ZD_StoreMode = packed), which the suite ties to the oracle at PPD = 256 / 512 (tests/test_gpu_parity.py) and —
PPD = 2048 PLT + rescale, BASELINE C3, now through the fused kernel — to the direct sum over every mode
(tests/test_gpu_direct_sum.py).  Whole planes of RVdoubleZel records out of a NaN-filled store: every displacement and velocity
component, the u16 indices, max_disp and density_variance."""
import ctypes as C

import numpy as np
import pytest

from conftest import WMAP

pytestmark = pytest.mark.gpu
FMT = "RVdoubleZel"


@pytest.fixture(scope="module")
def zd():
    import zeldovich_plt_amd.api as api
    api.load_library()
    return api


_STORE = {"t": None}
_EIG = {}


def _eig128(oracle):
    """the synthetic 128^3 eigenmode table (0.8 s of numpy per call: made once per module)"""
    if "t" not in _EIG:
        _EIG["t"] = oracle.synthetic_eigenmodes(128)
    return _EIG["t"]


def _planes(zd, ps, eig, n, zs, **kw):
    import torch
    dt = zd.RECORD_DTYPES[FMT]
    p = zd.make_params(n, icformat=FMT, qPLT=1, **kw)
    if p.stream_factor <= 0:
        free_b, _ = torch.cuda.mem_get_info()
        held = 0 if _STORE["t"] is None else _STORE["t"].numel()
        p.stream_factor = zd.load_library().zd_choose_stream_factor(C.byref(p), 1, int(free_b) + held - (24 << 30))
        assert p.stream_factor > 0
    plan = zd.Plan(p, ps, eig=eig)
    if _STORE["t"] is None or _STORE["t"].numel() < plan.exchange_bytes:
        _STORE["t"] = None
        torch.cuda.empty_cache()
        _STORE["t"] = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
    store = _STORE["t"]
    store.fill_(255)  # NaN bytes: an element nobody wrote shows up in the records
    out = torch.empty(n * n * dt.itemsize, dtype=torch.uint8, device="cuda")
    where = {plan.plane_z(ps_, lp): (ps_, lp) for ps_ in range(plan.passes) for lp in range(plan.local_planes)}
    res, last = {}, None
    for z in sorted(zs, key=lambda z: where[z]):
        pass_, lp = where[z]
        if last != pass_:
            plan.stage_z(pass_, store.data_ptr())
            plan.stage_y(store.data_ptr())
            last = pass_
        plan.stage_x(pass_, store.data_ptr(), lp, 1, out.data_ptr())  # one plane: the x kernel of the interleaved rows masks its pair
        torch.cuda.synchronize()
        res[z] = out.cpu().numpy().view(dt).reshape(n, n).copy()
    st = plan.stats()
    info = dict(R=plan.R, dv=st["density_variance"], maxd=np.array(st["max_disp"]), maxi=np.array(st["max_disp_index"]))
    plan.close()
    return res, info


@pytest.fixture(scope="module", autouse=True)
def _free_store():
    yield
    _STORE["t"] = None
    import torch
    torch.cuda.empty_cache()


CASES = [
    # (PPD, params, planes): R = 1 (no fold) and R = 2 (BASELINE C3: both residue passes, planes of the first and last groups of four)
    (1024, dict(qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0), [0, 3, 513, 1022]),
    (1024, dict(k_cutoff=2.0), [1, 766]),                       # pruned columns, kmax = 256: dead pairs inside live waves
    (1024, dict(corner_modes=1), [2, 1023]),                     # no sphere: every wave of every column live
    (2048, dict(qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0), [0, 1, 1030, 2047]),
    # z lines of 512 points (half a wave per line): PPD = 512 (BASELINE C2's size: also compared with the ORACLE record by record,
    # tests/test_gpu_baseline_regime.py::test_ppd512_plt_vs_oracle) and PPD = 1024 at R = 2
    (512, dict(qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0), [0, 2, 255, 511]),
    (1024, dict(stream_factor=2, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0), [1, 514, 1023]),
]


@pytest.mark.parametrize("n,kw,zs", CASES)
def test_fused_z_stage_equals_the_two_kernel_stage(zd, oracle, n, kw, zs):
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    eig = _eig128(oracle)
    a, ia = _planes(zd, ps, eig, n, zs, **kw)
    b, ib = _planes(zd, ps, eig, n, zs, store_mode="packed", **kw)
    import zeldovich_plt_amd.api as api
    rep = api.dispatch_report()
    assert any("launch_genz_t" in nm and cnt > 0 for (nm, _l), cnt in rep.items()), "the fused kernel did not run"
    worst = 0.0
    for z in zs:
        assert np.array_equal(a[z]["ijk"], b[z]["ijk"])
        scale = np.abs(b[z]["d"]).max()
        assert scale > 1e-3 and np.isfinite(a[z]["d"]).all() and np.isfinite(a[z]["v"]).all()
        worst = max(worst, np.abs(a[z]["d"] - b[z]["d"]).max() / scale, np.abs(a[z]["v"] - b[z]["v"]).max() / scale)
    print("PPD", n, kw, "R", ia["R"], "fused vs two kernels: max diff / max|q| =", worst)
    assert worst < 1e-12  # (the same arithmetic up to the association of the eigenmode blend: measured 6e-16)
    assert abs(ia["dv"] - ib["dv"]) <= 1e-13 * abs(ib["dv"])


def test_fused_z_stage_power_law_and_fixed_amplitudes(zd, oracle):
    """the PLAW instantiation and ZD_qPk_fix_to_mean through the fused kernel"""
    eig = _eig128(oracle)
    for ps in (zd.PowerSpectrum.from_powerlaw(-1.5, 720.0), zd.PowerSpectrum.from_file(WMAP, 720.0, fix_to_mean=1)):
        a, _ = _planes(zd, ps, eig, 1024, [5, 900])
        b, _ = _planes(zd, ps, eig, 1024, [5, 900], store_mode="packed")
        for z in (5, 900):
            scale = np.abs(b[z]["d"]).max()
            assert scale > 0 and np.abs(a[z]["d"] - b[z]["d"]).max() <= 1e-12 * scale and np.abs(a[z]["v"] - b[z]["v"]).max() <= 1e-12 * scale


def test_options_the_fused_kernel_does_not_take_keep_the_two_kernel_stage(zd, oracle):
    """CornerModes with ZD_k_cutoff = 2 (the kz = N/2 plane is live: the fused kernel never draws it), several ranks and
    ZD_StoreMode = packed run on plain rows as before.  (The one-mode filter IS taken by the fused kernel: the closed-form one-mode
    runs at PPD = 2048 PLT + rescale, tests/test_gpu_baseline_regime.py::test_large_plt_plane_waves_and_stream_invariance, go through it.)"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    eig = _eig128(oracle)
    import zeldovich_plt_amd.api as api

    def launches():
        return sum(cnt for (nm, _l), cnt in api.dispatch_report().items() if "launch_genz_t" in nm)

    n0 = launches()
    a, _ = _planes(zd, ps, eig, 1024, [7], corner_modes=1, k_cutoff=2.0)
    assert launches() == n0 and np.isfinite(a[7]["d"]).all()
    p = zd.make_params(1024, icformat=FMT, qPLT=1, stream_factor=1)
    plan = zd.Plan(p, ps, eig=eig, rank=0, nranks=2)  # two ranks: the exchange addresses plain rows
    plan.close()
    assert launches() == n0
    # ... and the one-mode filter through the fused kernel against the two-kernel stage: one live mode per pair, with its mirror zeroed
    for mode in ((3, 5, 7), (-9, 1, -2), (5, 2, 0), (17, 100, -400)):
        kw = dict(qonemode=1, one_mode=mode, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
        a, _ = _planes(zd, ps, eig, 1024, [4, 600], **kw)
        b, _ = _planes(zd, ps, eig, 1024, [4, 600], store_mode="packed", **kw)
        for z in (4, 600):
            scale = np.abs(b[z]["d"]).max()
            assert scale > 0 and np.abs(a[z]["d"] - b[z]["d"]).max() <= 1e-12 * scale and np.abs(a[z]["v"] - b[z]["v"]).max() <= 1e-12 * scale, mode
    assert launches() > n0
