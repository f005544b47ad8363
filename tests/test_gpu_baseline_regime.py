"""GPU parity in the regime the BASELINE configurations actually run in (VERDICT r1 "next" #1):

  * wmap1new.pow ends at k = 5.13 h/Mpc while PPD = 2048 / 4096 / 8192 at BoxSize 720 reach k_Nyquist = 8.9 / 17.9 / 35.7:
    most of their modes take P(k) from the LAST spline segment's extrapolation.  A small grid in a small box spans the
    same physical k, so the oracle can check that regime mode by mode and end to end;
  * the production generator arithmetic (LDS-table ln / exp / sincos / spline records, integer zero rule) on explicit
    mode lists at full-size wavenumbers;
  * full-size EXACT comparisons through the reference's oversampling invariant (README; src/zeldovich.cpp:350-356):
    PPD = 2N with ZD_k_cutoff = 2 sampled at even lattice sites == PPD = N with ZD_k_cutoff = 1, for 8192 <-> 4096
    (BASELINE configs C5 / C4 workloads) and 4096 <-> 2048 (C3 size);
  * C2 at its stated size: PPD = 512 with PLT eigenmodes (interpolated 128^3 table) against the oracle.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, WMAP, source_sha
from test_gpu_parity import TOL, _compare, _rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zd():
    import zeldovich_plt_amd.api as api
    api.load_library()
    return api


def _pair(zd, oracle, box, **kw):
    return zd.PowerSpectrum.from_file(WMAP, box, **kw), oracle.pk_from_file(WMAP, box, **kw)


# ---- (a) extrapolation regime, end to end -------------------------------------------------------------------------
@pytest.mark.parametrize("n,box,R", [(128, 22.5, 2), (256, 90.0, 4), (128, 11.25, 1)])  # k_Ny = 17.9, 8.9, 35.7
@pytest.mark.parametrize("store", ["auto", "reference"])
def test_za_extrapolated_pk_vs_oracle(zd, oracle, n, box, R, store):
    ps, opk = _pair(zd, oracle, box)
    assert np.pi * n / box > 1.7 * ps.pk.kmax  # most modes lie beyond the tabulated range
    _compare(zd, oracle, ps, opk, n, boxsize=box, stream_factor=R, store_mode=store)


@pytest.mark.parametrize("n,box,R,ppd_e", [(128, 22.5, 2, 48), (256, 90.0, 2, 128), (128, 11.25, 1, 64)])
@pytest.mark.parametrize("store", ["auto", "reference"])
def test_plt_rescale_extrapolated_pk_vs_oracle(zd, oracle, n, box, R, ppd_e, store):
    ps, opk = _pair(zd, oracle, box)
    eig = oracle.synthetic_eigenmodes(ppd_e)
    got, ref = _compare(zd, oracle, ps, opk, n, boxsize=box, stream_factor=R, store_mode=store, eig=eig, qPLT=1,
                        qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0, f_cluster=0.97)
    # a well-conditioned table: displacements stay of the order of the ZA ones (no near-singular k^2/(k.e) modes
    # carrying the comparison)
    za = oracle.run(oracle.make_params(n, numblock=2, boxsize=box), opk)
    assert np.abs(ref["max_disp"]).max() < 3.0 * np.abs(za["max_disp"]).max()


def test_plt_near_singular_eigenmodes_edge_case(zd, oracle):
    """round-1 synthetic table (e almost perpendicular to k at a few modes, k^2/(k.e) huge): still equal to the oracle"""
    ps, opk = _pair(zd, oracle, 720.0)
    eig = oracle.synthetic_eigenmodes(24, singular=True)
    _compare(zd, oracle, ps, opk, 64, eig=eig, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)


def test_fixed_amplitudes_and_smoothing_extrapolated(zd, oracle):
    ps, opk = _pair(zd, oracle, 22.5, fix_to_mean=1, Pk_smooth=0.05)
    _compare(zd, oracle, ps, opk, 128, boxsize=22.5, stream_factor=2)


def test_corner_modes_with_k_cutoff_uses_reference_arrays(zd, oracle):
    """with ZD_CornerModes = 1 and ZD_k_cutoff = 2 the Nyquist-plane modes survive as independent
    (non-Hermitian) draws; the packed stores assume Hermitian fields, so this configuration must run on the reference's
    arrays and equal the oracle"""
    ps, opk = _pair(zd, oracle, 720.0)
    plan = zd.Plan(zd.make_params(128, stream_factor=2, corner_modes=1, k_cutoff=2.0), ps)
    assert plan.narray == 2 and plan.plane_step == 1
    plan.close()
    _compare(zd, oracle, ps, opk, 128, stream_factor=2, corner_modes=1, k_cutoff=2.0)


# ---- (a') the production generator arithmetic on explicit modes -----------------------------------------------------
@pytest.mark.parametrize("n,box,fix", [(4096, 720.0, 0), (8192, 720.0, 0), (128, 22.5, 0), (4096, 720.0, 1), (1024, 90.0, 0)])
def test_table_generator_arithmetic_vs_oracle(zd, oracle, n, box, fix):
    """k_genf's LDS-table ln / exp / sincos / spline-record / Newton-reciprocal forms against cgauss<2> of the oracle
    (libm), mode by mode, at the wavenumbers of the full-size runs (|k| up to sqrt(3) * 2047 fundamentals)"""
    ps, opk = _pair(zd, oracle, box, fix_to_mean=fix)
    half = n // 2
    rng = np.random.default_rng(n + fix)
    m = 20000
    hi = min(half - 1, 4095)
    k = np.stack([rng.integers(-hi, hi + 1, m), rng.integers(0, min(half, 2048), m), rng.integers(-hi, hi + 1, m)], 1)
    # a band of modes just inside / outside the spherical cut, and small |k|
    k[:2000] = np.stack([rng.integers(-8, 9, 2000), rng.integers(0, 9, 2000), rng.integers(-8, 9, 2000)], 1)
    k2 = (k.astype(np.int64) ** 2).sum(1)
    p = zd.make_params(n, boxsize=box)
    op = oracle.make_params(n, boxsize=box)
    got, gik2 = zd.test_modes_table(p, ps, k)
    L = oracle.lib()
    r = (C.c_uint64 * 2)()
    D = (C.c_double * 2)()
    k2cut = op.nyquist ** 2
    ref = np.zeros(m, dtype=np.complex128)
    live = np.zeros(m, dtype=bool)
    for i in range(m):
        kx, ky, kz = (int(v) for v in k[i])
        if k2[i] == 0 or max(abs(kx), abs(ky), abs(kz)) == half or float(k2[i]) * op.fundamental ** 2 >= k2cut:
            continue
        L.zdo_mode_draw(C.byref(op), C.byref(opk), kx, ky, kz, r, D)
        ref[i] = D[0] + 1j * D[1]
        live[i] = True
    assert live.sum() > m // 3
    assert np.all(got[~live] == 0)
    err = np.abs(got[live] - ref[live]) / np.abs(ref[live])
    print("n", n, "max per-mode rel err", err.max(), "at |k|^2 =", k2[live][err.argmax()])
    assert err.max() < 1e-13
    ik2 = op.fundamental / (k2[live].astype(np.float64) * op.fundamental ** 2)
    assert np.abs(gik2[live] / ik2 - 1).max() < 4e-16


# ---- (b) full-size exact checks through the oversampling invariant ------------------------------------------------
# One block store for the runs of _planes, kept from test to test: hipMalloc + hipFree of a 140-236 GB store cost 5-10 s per run, more than
# the kernels of the one or two passes a test executes (the suite has ~60 such runs).  It grows when a plan needs more, and is
# released before every test of this module that does not go through _planes (those allocate inside the library) and at the end.
_STORE = {"t": None}
_STORE_USERS = {"test_oversampled_planes_exact_at_full_size", "test_radix7_oversampled_planes", "test_smooth_sizes_oversampled_planes",
                "test_plt_one_mode_at_every_composite_size", "test_z_lines_of_180", "test_density_one_mode_at_every_composite_size",
                "test_reference_and_packed_arrays_at_large_sizes", "test_density_planes_of_the_reference_arrays_at_4096",
                "test_pruned_columns_at_every_tile_width_from_poisoned_memory", "test_ppd16384_k_cutoff4_planes_equal_ppd4096",
                "test_large_plt_plane_waves_and_stream_invariance", "test_ppd6912_on_one_gpu_plane_waves"}


def _drop_store():
    if _STORE["t"] is not None:
        import torch
        _STORE["t"] = None
        torch.cuda.empty_cache()


def _store_tensor(nbytes):
    import torch
    if _STORE["t"] is None or _STORE["t"].numel() < nbytes:
        _drop_store()
        size = nbytes
        if nbytes > (64 << 30):  # a big store: take at once what the following big runs will ask for
            free_b, _ = torch.cuda.mem_get_info()
            size = max(nbytes, min(224 << 30, int(free_b) - (40 << 30)))  # (40 GB: the next plans' slabs and rings, kernel scratch)
        _STORE["t"] = torch.empty(size, dtype=torch.uint8, device="cuda")
    return _STORE["t"]


# tests of this module that allocate large buffers inside the library (the store must make room for them); the others — grids up to
# 512 — run beside the cached store (round 5: every drop costs the next user ~8 s of hipMalloc)
_MEMORY_HUNGRY = {"test_density_only_runs_at_large_sizes", "test_fnl_round_trip_identity_at_large_sizes",
                  "test_ppd2048_plt_store_and_stream_invariance", "test_plain_fma_build_passes_the_parity_suite"}


@pytest.fixture(autouse=True)
def _store_cache_guard(request):
    if request.node.originalname in _MEMORY_HUNGRY:
        _drop_store()
    yield


@pytest.fixture(autouse=True, scope="module")
def _store_cache_end():
    yield
    _drop_store()


def _planes(zd, ps, n, zs, stride=1, eig=None, fmt="Zeldovich", want_density=False, poison=False, **kw):
    """records of the z planes `zs` of a PPD = n run: only the passes that hold them are executed; stride > 1: only every
    stride-th lattice site of a plane leaves the GPU (a PPD = 16384 plane of records is 8.6 GB).  want_density (ZD_qdensity = 1):
    the float32 density planes too, as res["density"][z]"""
    import torch
    p = zd.make_params(n, icformat=fmt, **kw)
    if p.stream_factor <= 0:
        free_b, _ = torch.cuda.mem_get_info()
        held = 0 if _STORE["t"] is None else _STORE["t"].numel()  # (the cached store is this run's to use)
        p.stream_factor = zd.load_library().zd_choose_stream_factor(C.byref(p), 1, int(free_b) + held - (24 << 30))
        assert p.stream_factor > 0
    def make_plan():
        if not poison:
            return zd.Plan(p, ps, eig=eig)
        T = zd.load_testing_library()  # the -DZD_TESTING library with its rings (allocated by the plan) starting as NaN bytes, like the store below
        T.zd_test_poison(1)
        try:
            return zd.Plan(p, ps, eig=eig, testing=True)
        finally:
            T.zd_test_poison(0)

    try:
        plan = make_plan()
    except RuntimeError:
        if _STORE["t"] is None:
            raise
        _drop_store()  # the plan's own buffers (folded-input slabs, rings: up to ~40 GB at PPD = 16384) did not fit beside the cached store
        plan = make_plan()
    dt = zd.RECORD_DTYPES[fmt]
    step = plan.plane_step
    out = torch.empty(step * n * n * dt.itemsize, dtype=torch.uint8, device="cuda")  # (before the store: it sizes itself by what is left)
    dens = torch.empty(step * n * n, dtype=torch.float32, device="cuda") if want_density else None
    store = _store_tensor(plan.exchange_bytes)
    store[:plan.exchange_bytes].fill_(0xFF)  # NaN bytes: a kernel that reads store elements no kernel wrote (pruned column tiles) shows up
    where = {}
    for ps_ in range(plan.passes):
        for lp in range(plan.local_planes):
            z = plan.plane_z(ps_, lp)
            if z in zs:
                where[z] = (ps_, lp)
    assert sorted(where) == sorted(zs)
    res = {}
    for z, (pass_, lp) in sorted(where.items(), key=lambda kv: kv[1]):
        plan.stage_z(pass_, store.data_ptr())
        plan.stage_y(store.data_ptr())
        first = lp // step * step
        plan.stage_x(pass_, store.data_ptr(), first, step, out.data_ptr(), d_density=None if dens is None else dens.data_ptr())
        torch.cuda.synchronize()
        sel = out.view(step, n, n, dt.itemsize)[lp - first, ::stride, ::stride].contiguous()
        res[z] = sel.cpu().numpy().view(dt).reshape(n // stride, n // stride).copy()
        if dens is not None:
            res.setdefault("density", {})[z] = dens.view(step, n, n)[lp - first, ::stride, ::stride].cpu().numpy().copy()
    info = dict(R=plan.R, passes=plan.passes, narray=plan.narray)
    plan.close()
    del store, out, dens
    return res, info


# (slow: 3456 <-> 6912 — the composite kernels at those sizes run in the direct-sum fixtures ppd3456_za / ppd6912_plt_rescale and the
# one-mode tests; 1000 / 2500: the convolution family keeps its link 2000 <-> 4000c)
_SLOW = pytest.mark.slow


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, pytest.param(3456, marks=_SLOW), pytest.param(1000, marks=_SLOW), 2000,
                               pytest.param(2500, marks=_SLOW)])
def test_oversampled_planes_exact_at_full_size(zd, n):
    """PPD = 2n, ZD_k_cutoff = 2 at even lattice sites == PPD = n, ZD_k_cutoff = 1 (8192 <-> 4096: BASELINE C5 / C4,
    4096 <-> 2048: C3 size; 6912 <-> 3456: the production Abacus size 6912 = 2^8 3^3 on the composite-transform kernels;
    2000 <-> 1000: the any-PPD convolution kernels, zd_kernels_any.hip; 4000 (= 32 * 125: radix-5 composite kernels, round 3) <-> 2000
    (convolution kernels): two transform families against each other; 5000 <-> 2500: the convolution engines of 16384 and 8192 points,
    the former reached by no other test),
    records compared exactly (1e-13 of the field maximum) on planes of different passes.
    The links 2048 <-> 1024 <-> 512 <-> 256 close the chain: PPD = 512 and 256 are compared with the ORACLE record by record
    on the same default store (test_ppd512_za_default_store_vs_oracle, test_za_extrapolated_pk_vs_oracle), so every BASELINE
    size is tied to an oracle-checked run through exact links of HIP runs at different sizes (different z / y / x kernels
    and stream factors at each size)"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    zs = [5, n // 2 + 3, n - 2] if n < 2500 else ([5, n // 2 + 3] if n == 4096 else [n // 2 + 3])  # (planes of different passes; fewer at the big sizes: suite time)
    # (PPD = 2000 on the convolution kernels at R = 2: its one-pass store is 259 GB — larger than the store the tests of this module
    # share, and every test after it then paid an 8 s re-allocation; ZD_RUN_SLOW keeps the one-pass run)
    lo, ilo = _planes(zd, ps, n, zs, **(dict(stream_factor=2) if n == 2000 and not os.environ.get("ZD_RUN_SLOW") else {}))
    hi, ihi = _planes(zd, ps, 2 * n, [2 * z for z in zs], k_cutoff=2.0)
    print("PPD", n, ilo, "PPD", 2 * n, ihi)
    for z in zs:
        a, b = lo[z], hi[2 * z][::2, ::2]
        assert np.array_equal(a["ijk"][..., 0], np.full((n, n), z))
        assert np.array_equal(b["ijk"][..., 0], np.full((n, n), 2 * z))
        assert np.array_equal(2 * a["ijk"][..., 1:].astype(np.int64), b["ijk"][..., 1:].astype(np.int64))
        scale = np.abs(a["d"]).max()
        assert scale > 0
        err = np.abs(a["d"] - b["d"]).max() / scale
        print("  z", z, "max |d(2n, even sites) - d(n)| / max|d| =", err)
        assert err < (1e-13 if n in (2048, 4096) else 1e-12)  # composite transforms: 27-term outer sums; 1000 / 2000: convolutions


# (one link per family and chain end every time — 224 (oracle-anchored start), the top links 3584, 2688, 2240, 3136, 4320; the links in
# between are `slow`: their sizes still run the one-mode PLT / density tests below and the dispatch-coverage check)
@pytest.mark.parametrize("n", [224, pytest.param(448, marks=_SLOW), pytest.param(896, marks=_SLOW), pytest.param(1792, marks=_SLOW), 3584,
                               336, pytest.param(672, marks=_SLOW), pytest.param(1344, marks=_SLOW), 2688, 560,
                               pytest.param(1120, marks=_SLOW), 2240, 784, pytest.param(1568, marks=_SLOW), 3136,
                               pytest.param(2160, marks=_SLOW), 4320])
def test_radix7_oversampled_planes(zd, n):
    """Every composite grid with a radix-7 outer transform (round 4: Q = 7, 21, 35, 49 and Q = 135; csrc/zd_kernels_np2.hip NP2_SIZES)
    runs at its own size: PPD = 2n with ZD_k_cutoff = 2 at even sites == PPD = n.  The chains start from a run checked elsewhere:
    224 against the oracle (test_gpu_parity.py test_radix7_ppd_vs_oracle), 336 / 560 / 784 / 2160 (P = 16: no composite y / x
    kernels) on the convolution kernels of zd_kernels_any.hip — two transform families against each other.
    Q = 7: 224 <-> 448 <-> 896 <-> 1792 <-> 3584 <-> 7168;  Q = 21: 336c <-> 672 <-> 1344 <-> 2688 <-> 5376;
    Q = 35: 560c <-> 1120 <-> 2240 <-> 4480;  Q = 49: 784c <-> 1568 <-> 3136 <-> 6272;  Q = 135: 2160c <-> 4320 <-> 8640 (= 64 * 135: beyond 8192, y tiles one column wide)."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    zs = [5, n // 2 + 3, n - 2] if n < 3000 else [5, n // 2 + 3]
    comp = n not in (336, 560, 784, 2160)
    lo, ilo = _planes(zd, ps, n, zs, **(dict(stream_factor=2) if comp and n <= 2688 else {}))  # R = 2: the longest z lines one GPU holds
    hi, ihi = _planes(zd, ps, 2 * n, [2 * z for z in zs], k_cutoff=2.0)
    print("PPD", n, ilo, "PPD", 2 * n, ihi)
    assert ihi["narray"] == 3 and (ilo["narray"] == 3) == comp  # field store of the composite kernels / reference arrays of the convolutions
    for z in zs:
        a, b = lo[z], hi[2 * z][::2, ::2]
        assert np.array_equal(b["ijk"][..., 0], np.full((n, n), 2 * z))
        assert np.array_equal(2 * a["ijk"][..., 1:].astype(np.int64), b["ijk"][..., 1:].astype(np.int64))
        scale = np.abs(a["d"]).max()
        assert scale > 0
        err = np.abs(a["d"] - b["d"]).max() / scale
        print("  z", z, "max |d(2n, even sites) - d(n)| / max|d| =", err)
        assert err < 1e-12


# (per family Q = 3, 9, 5, 15, 25, 45, 75, 125: the first link and the top link every time, the ones in between `slow` —
# except 1920: the only run with z lines of 1920 = 128 * 15, which the dispatch-coverage check wants)
@pytest.mark.parametrize("n", [192, pytest.param(384, marks=_SLOW), pytest.param(768, marks=_SLOW), 1536, 1152, 2304, 320,
                               pytest.param(1280, marks=_SLOW), 2560, 960, 1920, 3840, 1600, 3200, 720,
                               pytest.param(1440, marks=_SLOW), 2880, 1200, 2400, 4000])
def test_smooth_sizes_oversampled_planes(zd, n):
    """The 3- and 5-smooth composite grids of rounds 2 and 3 that had no run AT their size in the suite (found with zd_dispatch_report:
    384, 640, 768, 1440, 1536, 1920, 2304, 2400, 2560, 2880, 3072, 3200, 3840, 4608, 4800, 5120, 5760, 6400, 7680, 8000 — their
    (P, Q) engines were covered by test_fft_lines_*, their y / x / z kernels were not): PPD = 2n with ZD_k_cutoff = 2 at even sites
    == PPD = n, as test_radix7_oversampled_planes.  The lower ends hang on runs checked elsewhere: 192, 320 against the oracle, 960,
    1152, 1280, 1600, 4000 through links of test_gpu_parity.py / test_oversampled_planes_exact_at_full_size, 720 and 1200 on the
    convolution kernels (P = 16)."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    zs = [5, n // 2 + 3, n - 2] if n < 3000 else [5, n // 2 + 3]
    comp = n not in (720, 1200)
    lo, ilo = _planes(zd, ps, n, zs)
    hi, ihi = _planes(zd, ps, 2 * n, [2 * z for z in zs], k_cutoff=2.0)
    print("PPD", n, ilo, "PPD", 2 * n, ihi)
    assert ihi["narray"] == 3 and (ilo["narray"] == 3) == comp
    for z in zs:
        a, b = lo[z], hi[2 * z][::2, ::2]
        assert np.array_equal(b["ijk"][..., 0], np.full((n, n), 2 * z))
        assert np.array_equal(2 * a["ijk"][..., 1:].astype(np.int64), b["ijk"][..., 1:].astype(np.int64))
        scale = np.abs(a["d"]).max()
        assert scale > 0
        err = np.abs(a["d"] - b["d"]).max() / scale
        print("  z", z, "max |d(2n, even sites) - d(n)| / max|d| =", err)
        assert err < 1e-12


def test_z_lines_of_180(zd):
    """z lines of 180 = 4 * 45 (the one 4 * Q form of launch_zfft_fields_np2 no other test reaches; PPD = 8640 takes it at R = 48):
    PPD = 1440 at R = 8 against R = 2 (z lines of 720 = 16 * 45) on sample planes"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    n, zs = 1440, [3, 725, 1438]
    a, ia = _planes(zd, ps, n, zs, stream_factor=2)
    b, ib = _planes(zd, ps, n, zs, stream_factor=8)
    assert ia["R"] == 2 and ib["R"] == 8 and ia["narray"] == ib["narray"] == 3
    for z in zs:
        assert np.array_equal(a[z]["ijk"], b[z]["ijk"])
        assert np.abs(a[z]["d"]).max() > 0
        assert np.abs(a[z]["d"] - b[z]["d"]).max() <= 1e-12 * np.abs(a[z]["d"]).max(), z


def _np2_sizes():
    """the (P, Q) of csrc/zd_kernels_np2.hip NP2_SIZES = every PPD the composite kernels are instantiated for"""
    import re
    src = open(os.path.join(ROOT, "zeldovich_plt_amd", "csrc", "zd_kernels_np2.hip")).read()
    body = re.search(r"#define NP2_SIZES\(X\)(.*?)\nint launch_yfft_fields_np2", src, re.S).group(1)
    return sorted(int(p_) * int(q) for p_, q, _w in re.findall(r"X\((\d+), (\d+), (\d+)\)", body))


@pytest.mark.parametrize("n", [n for n in _np2_sizes() if n <= 8192])
def test_plt_one_mode_at_every_composite_size(zd, oracle, n):
    """PLT + rescale on EVERY composite grid the library has kernels for (the PLT field store's x kernels k_xfft_q3<.., true> /
    k_xfft_seq_q<.., true> are separate instantiations per size; until round 4 only 96 ... 288, 160, 3456 and 6912 ran with PLT):
    a one-mode run against the closed form built from the ORACLE's per-mode pieces, every displacement and velocity component of a
    plane at every 8th site (see test_large_plt_plane_waves_and_stream_invariance), at the stream factor the library chooses."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    opk = oracle.pk_from_file(WMAP, 720.0)
    eig = oracle.synthetic_eigenmodes(32)
    z = n // 2 + 3
    fc, ztar, zini = 0.97, 5.0, 49.0
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=ztar, z_initial=zini, f_cluster=fc, fmt="RVdoubleZel", eig=eig)
    op = oracle.make_params(n, qPLT=1, qPLTrescale=1, PLT_target_z=ztar, z_initial=zini, f_cluster=fc)
    L = oracle.lib()
    fund = 2 * np.pi / 720.0
    mode = (-(n // 5) - 1, n // 7 + 2, n // 3 + 1)
    yy, xx = np.meshgrid(np.arange(0, n, 8), np.arange(0, n, 8), indexing="ij")
    got, info = _planes(zd, ps, n, [z], stride=8, qonemode=1, one_mode=mode, **kw)
    assert info["narray"] == 3  # the PLT field store of the composite kernels
    r, D, e = (C.c_uint64 * 2)(), (C.c_double * 2)(), (C.c_double * 4)()
    L.zdo_mode_draw(C.byref(op), C.byref(opk), mode[0], mode[1], mode[2], r, D)
    L.zdo_get_eigenmode(eig.ctypes.data, eig.shape[0], mode[0], mode[1], mode[2], n, 1, e)
    f = (np.sqrt(1 + 24 * e[3] * fc) - 1) / 4
    target_f = (np.sqrt(1 + 24 * fc) - 1) / 4
    rescale = ((1 / (1 + ztar)) / (1 / (1 + zini))) ** (target_f - f)
    k2 = sum(m * m for m in mode) * fund * fund
    t = 2 * np.pi * ((mode[0] * xx + mode[1] * yy + mode[2] * z) % n) / n
    wave = -2.0 * (D[0] * np.sin(t) + D[1] * np.cos(t))
    rec = got[z]
    scale = max(abs(rescale * e[j] * fund / k2) for j in range(3)) * np.abs(wave).max()
    assert scale > 0
    for j in range(3):
        want = rescale * e[j] * fund / k2 * wave
        assert np.abs(rec["d"][..., 2 - j] - want).max() <= 1e-12 * scale, (mode, j)
        assert np.abs(rec["v"][..., 2 - j] - f * want).max() <= 1e-12 * f * scale, (mode, j)


@pytest.mark.parametrize("n", _np2_sizes())
def test_density_one_mode_at_every_composite_size(zd, oracle, n):
    """ZA with ZD_qdensity = 1 on EVERY composite grid (six-field store: generator kind ZAFD, the density call of the y kernel and
    k_xdens_q<P, 16, Q> are instantiated per size; round 4 tested them at 96 ... 480 only): a one-mode run against the closed forms
    q_j(x) = -2 (k_j fund / k^2) (Re D sin t + Im D cos t), v = vnorm q and delta(x) = 2 (Re D cos t - Im D sin t) = -div q, with D(k)
    from the oracle's per-mode draw; the density plane is float32 (src/output.cpp:217-224).  (The small sizes of the list also run
    against the oracle with random fields: test_density_on_composite_grids_vs_oracle — which pins these closed forms.)"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    opk = oracle.pk_from_file(WMAP, 720.0)
    fc = 0.93
    op = oracle.make_params(n, f_cluster=fc)
    fund = 2 * np.pi / 720.0
    vnorm = (np.sqrt(1 + 24 * fc) - 1) / 4
    z = n // 2 + 3
    mode = (n // 7 + 1, n // 5 + 2, -(n // 3) + 1)  # (ky >= 0: the mode loop runs over the half space, src/zeldovich.cpp:333-340)
    yy, xx = np.meshgrid(np.arange(0, n, 8), np.arange(0, n, 8), indexing="ij")
    got, info = _planes(zd, ps, n, [z], stride=8, fmt="RVdoubleZel", f_cluster=fc, qonemode=1, one_mode=mode, qdensity=1, want_density=True)
    assert info["narray"] == 3  # the six-field store of the composite kernels, not the reference arrays of the convolution path
    r, D = (C.c_uint64 * 2)(), (C.c_double * 2)()
    oracle.lib().zdo_mode_draw(C.byref(op), C.byref(opk), mode[0], mode[1], mode[2], r, D)
    k2 = sum(m * m for m in mode) * fund * fund
    t = 2 * np.pi * ((mode[0] * xx + mode[1] * yy + mode[2] * z) % n) / n
    wave = -2.0 * (D[0] * np.sin(t) + D[1] * np.cos(t))
    scale = max(abs(m) for m in mode) * fund / k2 * np.abs(wave).max()
    rec = got[z]
    for j in range(3):
        want = mode[j] * fund / k2 * wave
        assert np.abs(rec["d"][..., 2 - j] - want).max() <= 1e-12 * scale, (mode, j)
        assert np.abs(rec["v"][..., 2 - j] - vnorm * want).max() <= 1e-12 * vnorm * scale, (mode, j)
    delta = 2.0 * (D[0] * np.cos(t) - D[1] * np.sin(t))
    assert np.abs(delta).max() > 0
    assert np.abs(got["density"][z] - delta).max() <= 2e-7 * np.abs(delta).max()  # float32 planes


@pytest.mark.parametrize("n,kc,base,other", [
    (512, 1.0, dict(plt=True), dict(plt=True, store_mode="reference")),                     # k_xfft<512,16,4,.>
    (2048, 1.0, dict(), dict(store_mode="reference")),                                     # k_xfft<2048,16,2,.>, k_yfft<2048>, k_zfft<1024>
    (2048, 1.0, dict(), dict(store_mode="packed")),                                        # ZA pairs: k_xfft<2048,16,3,.>
    (4096, 1.0, dict(), dict(store_mode="reference", stream_factor=16)),                   # the reference's two arrays at the headline size
    (4096, 1.0, dict(), dict(store_mode="packed", stream_factor=16)),
    (4096, 1.0, dict(plt=True), dict(plt=True, store_mode="reference", stream_factor=32)),  # four arrays at 4096
    (8192, 2.0, dict(), dict(store_mode="reference", stream_factor=128)),                  # k_yfft<8192>, k_xfft<8192,16,2,1>
])
def test_reference_and_packed_arrays_at_large_sizes(zd, oracle, n, kc, base, other):
    """The reference's own block-array layout (`ZD_StoreMode = reference`: what ZD_qdensity, ZD_f_NL and the any-PPD path use) and the
    round-1 packings at the sizes of BASELINE's configurations: the launchers launch_zfft_t / launch_yfft_t / launch_xfft_t<N, 16, NA>
    had no run above PPD = 1024 except NA = 4 at 2048 (zd_dispatch_report).  A random plane must equal the default store's (field
    stores / PLT3), which the oversampling chain and the direct sums tie to the oracle."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    eig = oracle.synthetic_eigenmodes(128)
    z = n // 2 + 3

    def run(kw):
        kw = dict(kw)
        e = None
        if kw.pop("plt", False):
            e = eig
            kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
        return _planes(zd, ps, n, [z], stride=4, fmt="RVdoubleZel", eig=e, k_cutoff=kc, **kw)

    a, ia = run(base)
    b, ib = run(other)
    print(ia, ib)
    assert ia["narray"] == 3 and ib["narray"] == (4 if other.get("plt") else (3 if other["store_mode"] == "packed" else 2))
    assert np.array_equal(a[z]["ijk"], b[z]["ijk"])
    for f_ in ("d", "v"):
        assert np.abs(a[z][f_]).max() > 0
        assert np.abs(a[z][f_] - b[z][f_]).max() <= 1e-12 * np.abs(a[z][f_]).max(), f_


@pytest.mark.parametrize("n,kc", [(4096, 1.0), (8192, 2.0)])
def test_density_planes_of_the_reference_arrays_at_4096(zd, n, kc):
    """ZD_qdensity at the headline size (and at 8192, BASELINE C5's grid): the reference's two arrays + the density array (ZD_qdensity = 1:
    k_xfft<N, 16, 2, .> with the density epilogue) and the density array alone (ZD_qdensity = 2: k_xfft<N, 16, 1, .>) must write the same
    float32 density plane, and the records of the former must equal the default store's"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    z = n // 2 + 3
    base, _ = _planes(zd, ps, n, [z], stride=4, fmt="RVZel", k_cutoff=kc)
    a, ia = _planes(zd, ps, n, [z], stride=4, fmt="RVZel", qdensity=1, want_density=True, k_cutoff=kc)
    b, ib = _planes(zd, ps, n, [z], stride=4, fmt="RVZel", qdensity=2, want_density=True, k_cutoff=kc)
    assert ia["narray"] == 2 and ib["narray"] == 1
    for f_ in ("d", "v"):
        assert np.abs(base[z][f_]).max() > 0
        assert np.abs(a[z][f_] - base[z][f_]).max() <= 1e-6 * np.abs(base[z][f_]).max()  # (RVZel: float32 records)
    da, db = a["density"][z], b["density"][z]
    assert np.abs(da).max() > 0 and np.abs(da - db).max() <= 2e-7 * np.abs(da).max()


@pytest.mark.parametrize("n", [512, pytest.param(1024, marks=_SLOW), pytest.param(2048, marks=_SLOW)])
def test_density_only_runs_at_large_sizes(zd, n):
    """ZD_qdensity = 2 (one array, no records: launch_xfft_t<N, 16, 1, .>) at the large powers of two: the variance of the density
    planes it writes must equal the generator's sum of |D|^2 of the default ZA run (Parseval; src/output.cpp:225-228 sums the planes)"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    a = zd.generate(zd.make_params(n, icformat="RVZel"), ps, collect=False)
    b = zd.generate(zd.make_params(n, icformat="RVZel", qdensity=2), ps, collect=False)
    assert a["density_variance"] > 0
    assert abs(a["density_variance"] - b["density_variance"]) <= 2e-6 * a["density_variance"]  # float32 planes summed in double


@pytest.mark.parametrize("n,R", [(256, 2), (1024, 4), pytest.param(2048, 8, marks=_SLOW)])
def test_fnl_round_trip_identity_at_large_sizes(zd, n, R):
    """The phi round of ZD_f_NL (inverse z / y / x transforms of phi = D / M, phi + f_NL phi^2, forward x / y / z transforms, D = phi M:
    k_xphi / k_yfwd / k_zfwd, launch_fnl_t<N>) at sizes the oracle cannot run: with f_NL = 1e-300 the nonlinear term vanishes and
    the second pass must reproduce the ordinary run — every record of sample planes to 1e-10 (src/zeldovich.cpp:699-790, 945-960).
    PPD = 2048 is the largest size whose phi field (137 GB) fits one GPU."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    zs = [3, n // 2 + 1]
    kw = dict(fmt="RVdoubleZel", stream_factor=R, n_s=0.96, Omega_M=0.31)
    a, ia = _planes(zd, ps, n, zs, stride=2, store_mode="reference", **kw)
    b, ib = _planes(zd, ps, n, zs, stride=2, f_NL=1e-300, **kw)
    assert ia["narray"] == ib["narray"] == 2
    for z in zs:
        for f_ in ("d", "v"):
            assert np.abs(a[z][f_]).max() > 0
            assert np.abs(a[z][f_] - b[z][f_]).max() <= 1e-10 * np.abs(a[z][f_]).max(), (z, f_)


@pytest.mark.parametrize("n,kc,kw", [
    (1024, 2.0, dict()), (2048, 2.0, dict()), (4096, 2.0, dict()), (8192, 2.0, dict(stream_factor=128)), (16384, 4.0, dict(stream_factor=256)),
    (2048, 2.0, dict(plt=True)), (4096, 2.0, dict(plt=True, stream_factor=64)), (8192, 2.0, dict(plt=True, stream_factor=256)),
    (3456, 2.0, dict(stream_factor=24)), (6912, 2.0, dict(stream_factor=48)), (7168, 2.0, dict(stream_factor=64)), (8640, 2.0, dict(stream_factor=80)),
    (3456, 2.0, dict(stream_factor=24, qdensity=1)), (6912, 2.0, dict(plt=True, stream_factor=64)),
])
def test_pruned_columns_at_every_tile_width_from_poisoned_memory(zd, oracle, n, kc, kw):
    """ZD_k_cutoff > 1 leaves column tiles of the stores and of the y -> x rings unwritten; which tiles depends on the tile widths, and
    those follow the grid size (32 columns at PPD = 128 ... 1 at 16384 / 8640; 8 / 4 / 2 on the composite grids).  The small sizes are
    swept by test_poison_sweep_of_the_stores; here one short pass of every width class at full size runs in the -DZD_TESTING library with
    its rings and the block store starting as NaN bytes: a plane that is not finite shows a kernel reading what no kernel wrote."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    kw = dict(kw)
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(32)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
    z = n // 2 + 6
    dens = bool(kw.get("qdensity"))
    got, info = _planes(zd, ps, n, [z], stride=2 if n < 8000 else 4, fmt="RVdoubleZel", eig=eig, k_cutoff=kc, poison=True, want_density=dens, **kw)
    assert info["narray"] == 3
    for f_ in ("d", "v"):
        assert np.isfinite(got[z][f_]).all(), f_
        assert np.abs(got[z][f_]).max() > 0
    if dens:
        assert np.isfinite(got["density"][z]).all()


def test_ppd16384_k_cutoff4_planes_equal_ppd4096(zd):
    """PPD = 16384 (beyond the 8192 of BASELINE C5; MAX_PPD = 65536, include/zeldovich.h:34) with ZD_k_cutoff = 4 at every
    fourth lattice site == PPD = 4096: one plane, records compared exactly"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    n, z = 4096, 4096 // 2 + 3
    lo, ilo = _planes(zd, ps, n, [z])
    hi, ihi = _planes(zd, ps, 4 * n, [4 * z], stride=4, k_cutoff=4.0)
    print("PPD", n, ilo, "PPD", 4 * n, ihi)
    a, b = lo[z], hi[4 * z]
    assert np.array_equal(b["ijk"][..., 0], np.full((n, n), 4 * z))
    assert np.array_equal(4 * a["ijk"][..., 1:].astype(np.int64), b["ijk"][..., 1:].astype(np.int64))
    err = np.abs(a["d"] - b["d"]).max() / np.abs(a["d"]).max()
    print("  max |d(16384, every 4th site) - d(4096)| / max|d| =", err)
    assert err < 1e-13


@pytest.mark.parametrize("n,kc,Ra,Rb,modes", [
    (2048, 1.0, 2, 4, [(3, 5, -7), (-401, 577, 600), (0, 2, 0)]),          # BASELINE C3: packed PLT arrays (k_genf PLTN / k_zfft / k_yfft / k_xfft)
    (4096, 1.0, 16, 32, [(3, 5, -7), (-1001, 777, 1200), (0, 2, 0)]),    # PLT at the headline grid: k_genf PLTN + k_eig_lines at 4096 / k_zfft<256|128> / k_yfft<4096> / k_xfft_seq_plt<4096,16,true>
    (8192, 2.0, 64, 128, [(3, 5, -7), (-1001, 777, 1200), (0, 2, 0)]),   # k_xfft_two<8192, PLT>
    # (6912 at k_cutoff = 4, R = 8 / 16, until round 4: PLT at 6912 now has the direct sum over every mode at k_cutoff = 1 —
    #  tests/test_gpu_direct_sum.py[ppd6912_plt_rescale] — besides the one-mode, poisoned-memory and R = 64 runs)
    (3456, 2.0, 4, 8, [(-3, 5, 7), (401, 377, -500)]),                   # composite kernels, three lines per workgroup
    (6912, 1.0, 64, None, [(-2001, 1777, 1200)]),                        # production Abacus on ONE GPU: z lines of 108 = 4 * 27
    (3584, 2.0, 8, 16, [(-401, 377, 500)]),                              # radix-7 composite kernels (3584 = 512 * 7) with the PLT field store
])
def test_large_plt_plane_waves_and_stream_invariance(zd, oracle, n, kc, Ra, Rb, modes):
    """PLT + rescale at the sizes that only the PLT FIELD store serves — PPD = 8192 (the y pass at 8192 and the x pass in two
    launches, `k_xfft_two`) and PPD = 2^a 3^b (6912 = the production Abacus grid, 3456) — checked at full size.  (There is no
    oversampling invariant with PLT: the eigenmode of a physical k depends on the particle lattice,
    src/zeldovich.cpp:154-227.)
    (i) one-mode runs against the closed form built from the ORACLE's per-mode pieces (its draw D(k) and get_eigenmode at
        this ppd): q_j(x) = -2 s_j (Re D sin t + Im D cos t), t = 2 pi k.x / N, s = rescale e fund / k^2, v = f q
        (src/zeldovich.cpp:403-452) — every component of displacement and velocity of a plane;
    (ii) a full random plane is independent of the stream factor."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    opk = oracle.pk_from_file(WMAP, 720.0)
    eig = oracle.synthetic_eigenmodes(128)
    z = n // 2 + 3
    fc, ztar, zini = 0.97, 5.0, 49.0
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=ztar, z_initial=zini, f_cluster=fc, fmt="RVdoubleZel", eig=eig, k_cutoff=kc)
    op = oracle.make_params(n, k_cutoff=kc, qPLT=1, qPLTrescale=1, PLT_target_z=ztar, z_initial=zini, f_cluster=fc)
    L = oracle.lib()
    fund = 2 * np.pi / 720.0
    yy, xx = np.meshgrid(np.arange(0, n, 16), np.arange(0, n, 16), indexing="ij")
    for mode in modes:
        got, _ = _planes(zd, ps, n, [z], stride=16, stream_factor=Ra, qonemode=1, one_mode=mode, **kw)
        r, D, e = (C.c_uint64 * 2)(), (C.c_double * 2)(), (C.c_double * 4)()
        L.zdo_mode_draw(C.byref(op), C.byref(opk), mode[0], mode[1], mode[2], r, D)
        L.zdo_get_eigenmode(eig.ctypes.data, eig.shape[0], mode[0], mode[1], mode[2], n, 1, e)
        f = (np.sqrt(1 + 24 * e[3] * fc) - 1) / 4
        target_f = (np.sqrt(1 + 24 * fc) - 1) / 4
        rescale = ((1 / (1 + ztar)) / (1 / (1 + zini))) ** (target_f - f)
        k2 = sum(m * m for m in mode) * fund * fund
        t = 2 * np.pi * ((mode[0] * xx + mode[1] * yy + mode[2] * z) % n) / n
        wave = -2.0 * (D[0] * np.sin(t) + D[1] * np.cos(t))
        rec = got[z]
        scale = max(abs(rescale * e[j] * fund / k2) for j in range(3)) * np.abs(wave).max()  # of the displacement vector
        assert scale > 0
        for j in range(3):  # records hold (qz, qy, qx): component j of this code's x, y, z order is column 2 - j
            want = rescale * e[j] * fund / k2 * wave
            assert np.abs(rec["d"][..., 2 - j] - want).max() <= 1e-12 * scale, (mode, j)
            assert np.abs(rec["v"][..., 2 - j] - f * want).max() <= 1e-12 * f * scale, (mode, j)
    if Rb is None:  # only one stream factor fits the GPU
        return
    a, ia = _planes(zd, ps, n, [z], stride=8, stream_factor=Ra, **kw)
    b, ib = _planes(zd, ps, n, [z], stride=8, stream_factor=Rb, **kw)
    print(ia, ib)
    for f_ in ("d", "v"):
        assert np.abs(a[z][f_]).max() > 0
        assert np.abs(a[z][f_] - b[z][f_]).max() <= 1e-12 * np.abs(a[z][f_]).max()


def test_ppd512_za_default_store_vs_oracle(zd, oracle):
    """PPD = 512 ZA on the DEFAULT store (field store, two z-residues per pass) against the oracle: 1.3e8 particles, every
    record — the anchor of the oversampling chain 8192 <-> 4096 <-> 2048 <-> 1024 <-> 512 (test_oversampled_planes_exact_at_full_size)"""
    ps, opk = _pair(zd, oracle, 720.0)
    got, _ = _compare(zd, oracle, ps, opk, 512)
    assert got["stream_factor"] >= 2  # the field store carries two residues per pass


@pytest.mark.parametrize("n,R,modes", [
    (6912, 64, [(-2001, 1777, 1200), (5, 3, -7)]),   # R given: z lines of 108 = 4 * 27 (the only factor a power of two offers)
    (6912, 0, [(-2001, 1777, 1200)]),                # R chosen: 36 (z lines of 192 = 64 * 3, 18 passes) on a 288 GB GPU
    (4096, 8, [(-1001, 1177, 1200), (5, 3, -7), (0, 2, 0)]),   # the bench workload: k_genf / k_zfft_f<512> / k_yfft_f<4096> / k_xfft<4096,16,3,1>
])
def test_ppd6912_on_one_gpu_plane_waves(zd, oracle, n, R, modes):
    """PPD = 6912 = 2^8 3^3 at ZD_k_cutoff = 1 — the production Abacus grid — on ONE GPU: R = 64, z lines of 108 = 4 * 27, and the
    factor the library chooses, R = 36;
    and PPD = 4096 at ZD_k_cutoff = 1 (the headline workload, R = 8) through the general one-mode generator + the
    production z / y / x kernels.
    One-mode runs against the closed form q_j(x) = -2 (k_j fund / k^2) (Re D sin t + Im D cos t), v = vnorm q, with D(k) from
    the oracle's per-mode draw (one mode of each list lies beyond the last tabulated k = 5.13 h/Mpc = 588 fundamentals); the
    random-field path of the composite kernels is covered at PPD = 864 / 1728 (test_non_power_of_two_short_z_lines) and by
    6912 (k_cutoff = 2) <-> 3456."""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    opk = oracle.pk_from_file(WMAP, 720.0)
    z, fc = n // 2 + 5, 0.9
    op = oracle.make_params(n, f_cluster=fc)
    L = oracle.lib()
    fund = 2 * np.pi / 720.0
    vnorm = (np.sqrt(1 + 24 * fc) - 1) / 4
    yy, xx = np.meshgrid(np.arange(0, n, 16), np.arange(0, n, 16), indexing="ij")
    for mode in modes:
        got, info = _planes(zd, ps, n, [z], stride=16, fmt="RVdoubleZel", f_cluster=fc, qonemode=1, one_mode=mode,
                            stream_factor=R if n == 6912 else 0)
        assert info["R"] == (R if R else 36)
        r, D = (C.c_uint64 * 2)(), (C.c_double * 2)()
        L.zdo_mode_draw(C.byref(op), C.byref(opk), mode[0], mode[1], mode[2], r, D)
        k2 = sum(m * m for m in mode) * fund * fund
        t = 2 * np.pi * ((mode[0] * xx + mode[1] * yy + mode[2] * z) % n) / n
        wave = -2.0 * (D[0] * np.sin(t) + D[1] * np.cos(t))
        scale = max(abs(m) for m in mode) * fund / k2 * np.abs(wave).max()
        rec = got[z]
        for j in range(3):
            want = mode[j] * fund / k2 * wave
            assert np.abs(rec["d"][..., 2 - j] - want).max() <= 1e-12 * scale, (mode, j)
            assert np.abs(rec["v"][..., 2 - j] - vnorm * want).max() <= 1e-12 * vnorm * scale, (mode, j)


@pytest.mark.slow  # (15 s of 208 GB allocations; C3 is anchored by its direct-sum fixture and by tests/test_gpu_fused_z.py, the reference's
                   # four arrays at 2048 by the random-plane comparison above)
def test_ppd2048_plt_store_and_stream_invariance(zd, oracle):
    """C3 (PPD = 2048, PLT + rescale): the packed store and the reference's four arrays give the same reductions over all 8.6e9
    particles (size-independent property; the oracle cannot run that many)"""
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    eig = oracle.synthetic_eigenmodes(128)
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0, icformat="RVZel")
    a = zd.generate(zd.make_params(2048, **kw), ps, eig=eig, collect=False)
    b = zd.generate(zd.make_params(2048, store_mode="reference", **kw), ps, eig=eig, collect=False)
    for o in (b,):  # (R = 2 against R = 4: test_large_plt_plane_waves_and_stream_invariance[2048...], on a random plane)
        assert abs(a["density_variance"] - o["density_variance"]) <= 1e-11 * a["density_variance"]
        assert np.abs(a["max_disp"] - o["max_disp"]).max() <= 1e-11 * np.abs(a["max_disp"]).max()
    assert np.abs(a["max_disp"]).max() < 20.0  # well-conditioned eigenmodes: displacements O(1) Mpc/h, as for ZA


# ---- (c) BASELINE C2 at its stated size ---------------------------------------------------------------------------
def test_ppd512_plt_vs_oracle(zd, oracle):
    """PPD = 512 with PLT eigenmodes from a 128^3 table (trilinear interpolation, src/zeldovich.cpp:154-227) against
    the oracle: 1.3e8 particles, every record compared"""
    ps, opk = _pair(zd, oracle, 720.0)
    eig = oracle.synthetic_eigenmodes(128)
    _compare(zd, oracle, ps, opk, 512, eig=eig, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)


# ---- asm-FMA A/B -----------------------------------------------------------------------------------------------------
def test_plain_fma_build_passes_the_parity_suite():
    """the -DZD_NO_FMA_ASM library (plain fma() instead of the asm 3-address FMAs, `make nofma`) through a subset of the
    parity tests in a child process (ZD_LIB_PATH): generator arithmetic must not depend on the asm forms"""
    lib = os.path.join(ROOT, "zeldovich_plt_amd", "csrc", "build", "libzeldovich_hip_nofma.so")
    # round 3 ran this against a library older than the kernels it was meant to check (the make rule had been lost): a missing
    # or stale variant is a FAILURE.  Staleness by content, not mtime (snapshots do not keep mtimes): the Makefile records the
    # sha-256 of the sources next to every library it links
    assert os.path.exists(lib), "nofma variant not built (make -C zeldovich_plt_amd/csrc nofma; build() does it)"
    assert open(lib + ".srcsha").read().split()[0] == source_sha(), "libzeldovich_hip_nofma.so is older than the kernel sources"
    env = dict(os.environ, ZD_LIB_PATH=lib, ZD_TESTING_LIB_PATH=lib)  # (the variant carries the test hooks too)
    sel = "test_za_extrapolated_pk_vs_oracle or test_plt_rescale_extrapolated_pk_vs_oracle or test_table_generator_arithmetic_vs_oracle"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k", sel],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
