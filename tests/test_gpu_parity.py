"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeds.

Tolerances (BASELINE.json north_star): displacements / velocities <= 1e-10 relative to the field's
maximum (double precision); integer record fields exact; raw RNG draws bit-exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def zd():
    import zeldovich_plt_amd.api as api
    from conftest import OneGroupApi
    api.load_library()
    return OneGroupApi(api)


@pytest.fixture(scope="module")
def ps(zd, wmap_path):
    return zd.PowerSpectrum.from_file(wmap_path, 720.0)


@pytest.fixture(scope="module")
def opk(oracle, wmap_path):
    return oracle.pk_from_file(wmap_path, 720.0)


def test_draws_bit_exact(zd, oracle):
    """counter-addressed pcg64 draws == sequential reference stream"""
    import ctypes as C
    rng = np.random.default_rng(5)
    k = np.stack([rng.integers(-2047, 2048, 4000), rng.integers(0, 2048, 4000), rng.integers(-2047, 2048, 4000)], 1)
    k[:6] = [[3, 5, 7], [-3, 5, 7], [3, 5, -7], [-9, 1, -2], [1, 0, 2], [0, 2, 0]]
    got = zd.test_draws(12346, k)
    # golden vectors recorded from the reference
    gold = [(0x73da5750b0db04d6, 0xcd339d3f2f327ebd), (0x0a1dac63ed2b85b8, 0xbb7c44248c0563e1),
            (0x777373dcf4c2866b, 0xc3988c4a2ab1f703), (0xd513f641bab2a889, 0x88df5069d40b5329),
            (0x95e443207a26a8b2, 0x8e5348004a6e721b), (0xc658b8961a1051d7, 0x5f97cac5840856c6)]
    for i, (a, b) in enumerate(gold):
        assert int(got[i, 0]) == a and int(got[i, 1]) == b
    L = oracle.lib()
    p = oracle.make_params(4096)
    pk = oracle.pk_from_powerlaw(-1.0, 720.0)
    r = (C.c_uint64 * 2)()
    D = (C.c_double * 2)()
    for i in range(0, 4000, 7):
        L.zdo_mode_draw(C.byref(p), C.byref(pk), int(k[i, 0]), int(k[i, 1]), int(k[i, 2]), r, D)
        assert int(got[i, 0]) == r[0] and int(got[i, 1]) == r[1]


def test_mode_amplitudes(zd, oracle, ps, opk):
    """D(k) = sqrt(-P ln R) e^{2 pi i theta} on the device vs cgauss<2> in the oracle"""
    import ctypes as C
    n = 256
    rng = np.random.default_rng(6)
    k = np.stack([rng.integers(-127, 128, 3000), rng.integers(0, 128, 3000), rng.integers(-127, 128, 3000)], 1)
    p = zd.make_params(n)
    got = zd.test_modes(p, ps, k)
    op = oracle.make_params(n)
    L = oracle.lib()
    r = (C.c_uint64 * 2)()
    D = (C.c_double * 2)()
    ref = np.zeros(len(k), dtype=np.complex128)
    k2cut = op.nyquist ** 2
    for i in range(len(k)):
        kx, ky, kz = (int(v) for v in k[i])
        if max(abs(kx), abs(ky), abs(kz)) == n // 2 or (kx * kx + ky * ky + kz * kz) * op.fundamental ** 2 >= k2cut:
            continue
        L.zdo_mode_draw(C.byref(op), C.byref(opk), kx, ky, kz, r, D)
        ref[i] = D[0] + 1j * D[1]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("mode amplitude rel err", err)
    assert err < 1e-13


@pytest.mark.parametrize("n", [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("kind", [0, 1])
def test_fft_lines(zd, n, kind):
    """the register/LDS FFT engine vs numpy (unnormalised inverse DFT), both LDS layouts"""
    rng = np.random.default_rng(n + kind)
    lines = 64
    x = rng.standard_normal((lines, n)) + 1j * rng.standard_normal((lines, n))
    got = zd.test_fft(x, kind)
    ref = np.fft.ifft(x, axis=1) * n
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 5e-15, err


def _compare(zd, oracle, ps, opk, n, fmt="RVdoubleZel", eig=None, tie_ok=False, **kw):
    loopback = kw.pop("loopback", False)  # tests: the RCCL branch on the in-process emulation (zd_test_generate_loopback)
    p = zd.make_params(n, icformat=fmt, **kw)
    okw = dict(kw)
    okw.pop("stream_factor", None)
    okw.pop("store_mode", None)
    okw.pop("ngpu", None)
    okw.pop("exchange_planes", None)
    okw.pop("pass_groups", None)
    if "corner_modes" in okw:
        okw["CornerModes"] = okw.pop("corner_modes")
    op = oracle.make_params(n, numblock=okw.pop("numblock", 2), icformat=fmt, **okw)
    got = zd.generate(p, ps, eig=eig, loopback=loopback)
    ref = oracle.run(op, opk, eig=eig, eig_ppd=0 if eig is None else eig.shape[0],
                     want_density=bool(kw.get("qdensity", 0)))
    if ref["records"] is not None:
        g, r = got["records"], ref["records"]
        if "ijk" in g.dtype.names:
            assert np.array_equal(g["ijk"], r["ijk"])
        for f in ("d", "v"):
            if f in g.dtype.names:
                tol = TOL if g[f].dtype == np.float64 else 1e-6
                for c in range(3):
                    assert _rel(g[f][..., c], r[f][..., c]) < tol, (f, c, _rel(g[f][..., c], r[f][..., c]))
        if "d" in g.dtype.names and g["d"].dtype == np.float64 and got["records"] is not None:
            # output.cpp:190-193: max_disp[j] is the SIGNED displacement of largest magnitude, the first one in (z, y, x) order
            # on a tie (a strict > in the reference's loop).  Exact on the GPU's own records: value and lattice site
            for j in range(3):  # max_disp is in (x, y, z) order, records hold (qz, qy, qx)
                col = g["d"][..., 2 - j].ravel()
                first = int(np.argmax(np.abs(col)))  # numpy returns the first occurrence of the maximum
                if not np.any(col):
                    first = -1  # an identically zero field (e.g. a one-mode run whose mode is cut, or has ky < 0: never drawn)
                assert got["max_disp_index"][j] == first, (j, got["max_disp_index"][j], first)
                assert got["max_disp"][j] == col[first], (j, got["max_disp"][j], col[first])
        if tie_ok:  # a single plane wave: +max and -max agree to rounding, which of them is the larger (or the first of an exact
            # tie) depends on the last bit of the two FFTs — compare the magnitudes with the oracle
            assert _rel(np.abs(got["max_disp"]), np.abs(ref["max_disp"])) < TOL
        else:
            assert _rel(got["max_disp"], ref["max_disp"]) < TOL
    if ref["density"] is not None:
        assert _rel(got["density"], ref["density"]) < 1e-6
    assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]
    return got, ref


@pytest.mark.parametrize("n", [32, 64, 128])
def test_za_end_to_end(zd, oracle, ps, opk, n):
    got, ref = _compare(zd, oracle, ps, opk, n)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("R", [2, 4])
def test_za_residue_streaming(zd, oracle, ps, opk, R):
    """z-residue streaming (R passes) must reproduce the single-pass result"""
    _compare(zd, oracle, ps, opk, 128, stream_factor=R)


@pytest.mark.parametrize("fmt", ["RVZel", "Zeldovich", "ZelSimple"])
def test_record_formats(zd, oracle, ps, opk, fmt):
    _compare(zd, oracle, ps, opk, 64, fmt=fmt)


@pytest.mark.parametrize("ppd_e", [64, 24])  # exact-stride lookup and trilinear interpolation
def test_plt_end_to_end(zd, oracle, ps, opk, ppd_e):
    eig = oracle.synthetic_eigenmodes(ppd_e)
    _compare(zd, oracle, ps, opk, 64, eig=eig, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)


def test_plt_streaming(zd, oracle, ps, opk):
    eig = oracle.synthetic_eigenmodes(32)
    _compare(zd, oracle, ps, opk, 128, eig=eig, qPLT=1, stream_factor=2)


@pytest.mark.parametrize("n,kw", [
    (128, dict(stream_factor=2, k_cutoff=2.0)),                       # packed ZA pair + pruned tiles
    (128, dict(stream_factor=4, corner_modes=1)),                     # two passes of two residues, CornerModes
    (256, dict(stream_factor=8, fmt="RVZel")),                        # four passes, float records
    (128, dict(stream_factor=2, fmt="Zeldovich")),
    (64, dict(stream_factor=2, fmt="ZelSimple", k_cutoff=4.0)),
])
@pytest.mark.parametrize("store", ["fields", "packed"])
def test_packed_za_pairs_sweep(zd, oracle, ps, opk, n, kw, store):
    """ZA without ZD_qdensity and R >= 2: two z-residues per pass, density_variance from sum |D|^2.
    store = fields: the potentials E, Z of the half-space rows, zero columns not stored (PACK_ZAFIELD, the default);
    store = packed: the three arrays (qy + i qz)_r0 | (qy + i qz)_r1 | qx_r0 + i qx_r1 with Hermitian twins (PACK_ZAPAIR)"""
    kw = dict(kw, store_mode=store)
    fmt = kw.pop("fmt", "RVdoubleZel")
    plan = zd.Plan(zd.make_params(n, icformat=fmt, **kw), ps)
    assert plan.narray == 3 and plan.plane_step == 2
    ref_store = zd.Plan(zd.make_params(n, icformat=fmt, **dict(kw, store_mode="packed")), ps)
    if store == "fields":  # 4 half-space fields (= 2 arrays' worth) minus the zero columns, vs 3 arrays
        assert plan.exchange_bytes <= ref_store.exchange_bytes * 2 // 3
    ref_store.close()
    plan.close()
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("R,ppd_e,resc", [(1, 64, 1), (4, 24, 0), (2, 128, 1)])
@pytest.mark.parametrize("store", ["auto", "fields"])
def test_packed_plt_sweep(zd, oracle, ps, opk, R, ppd_e, resc, store):
    """PLT without ZD_qdensity: qx + i vx | qy + i qz | vy + i vz, any R; exact-stride and interpolated eigenmodes.
    store = fields: the opt-in PLT field store (the six sums X, Y, Z, fX, fY, fZ of the half-space rows; measured slower than
    the three packed arrays, which stay the default)"""
    eig = oracle.synthetic_eigenmodes(ppd_e)
    kw = dict(qPLT=1, qPLTrescale=resc, PLT_target_z=3.0, f_cluster=0.95, stream_factor=R, store_mode=store)
    plan = zd.Plan(zd.make_params(128, **kw), ps, eig=eig)
    assert plan.narray == 3 and plan.plane_step == 1 and plan.passes == R
    plan.close()
    _compare(zd, oracle, ps, opk, 128, eig=eig, **kw)


def test_packed_powerlaw_and_fixed_amplitudes(zd, oracle):
    """the table-driven generator on a power-law spectrum (no spline) and with ZD_qPk_fix_to_mean"""
    ps2 = zd.PowerSpectrum.from_powerlaw(-1.5, 720.0, fix_to_mean=1)
    opk2 = oracle.pk_from_powerlaw(-1.5, 720.0, fix_to_mean=1)
    _compare(zd, oracle, ps2, opk2, 64, stream_factor=2)
    ps3 = zd.PowerSpectrum.from_powerlaw(-2.0, 720.0, Pk_smooth=2.0)
    opk3 = oracle.pk_from_powerlaw(-2.0, 720.0, Pk_smooth=2.0)
    _compare(zd, oracle, ps3, opk3, 64)


def test_k_cutoff_and_density(zd, oracle, ps, opk):
    _compare(zd, oracle, ps, opk, 64, k_cutoff=2.0, qdensity=1)


def test_density_only(zd, oracle, ps, opk):
    _compare(zd, oracle, ps, opk, 64, qdensity=2)


def test_ppd8192_on_four_reference_arrays_is_refused_at_plan_creation(zd, oracle, ps):
    """PPD = 8192 with PLT and an option that needs the reference's arrays (ZD_qdensity, ZD_f_NL, ZD_StoreMode = reference): the x pass
    of four lines of 512 threads does not exist.  The plan must say so when it is created — not after the Z and y stages of the first
    pass (launch_xfft_t's own check) — and the stream-factor chooser must offer nothing."""
    import ctypes as C
    eig = oracle.synthetic_eigenmodes(16)
    for kw in (dict(qdensity=1), dict(store_mode="reference")):
        p = zd.make_params(8192, k_cutoff=2.0, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, stream_factor=256, **kw)
        with pytest.raises(RuntimeError):
            zd.Plan(p, ps, eig=eig)
        p.stream_factor = 0
        assert zd.load_library().zd_choose_stream_factor(C.byref(p), 1, 250 << 30) == -1
    plan = zd.Plan(zd.make_params(8192, k_cutoff=2.0, store_mode="reference", stream_factor=128), ps)  # two arrays: 1024 threads
    assert plan.narray == 2
    plan.close()


@pytest.mark.parametrize("kw", [dict(qdensity=2), dict(store_mode="packed"), dict(plt=True, store_mode="reference"), dict(plt=True),
                                dict(f_NL=2.0e4)])
def test_smallest_grid_every_store(zd, oracle, ps, opk, wmap_path, kw):
    """PPD = 32, the smallest grid of the dispatch tables (launch_xfft_t<32, 16, 1 | 3 | 4, .>, launch_fnl_t<32>: no other test ran
    them at this size): density only, the packed stores, PLT on the reference's four arrays, f_NL — against the oracle"""
    import ctypes as C
    kw = dict(kw)
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(16)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    if "f_NL" in kw:
        kw.update(n_s=0.96, Omega_M=0.31)
        opk = oracle.pk_from_file(wmap_path, 720.0)
        oracle.lib().zdo_pk_set_primordial(C.byref(opk), 0.96)
    _compare(zd, oracle, ps, opk, 32, eig=eig, **kw)


def test_one_mode_and_fixed_power(zd, oracle, wmap_path):
    ps2 = zd.PowerSpectrum.from_file(wmap_path, 720.0, fix_to_mean=1)
    opk2 = oracle.pk_from_file(wmap_path, 720.0, fix_to_mean=1)
    _compare(zd, oracle, ps2, opk2, 64, tie_ok=True, qonemode=1, one_mode=(3, 5, -7))
    _compare(zd, oracle, ps2, opk2, 64)


def test_oneslab(zd, oracle, ps, opk):
    p = zd.make_params(64, qoneslab=17)
    got = zd.generate(p, ps)
    assert got["planes_seen"] == [17]
    ref = oracle.run(oracle.make_params(64, qoneslab=17), opk)
    assert _rel(got["records"]["d"][17], ref["records"]["d"][17]) < TOL
    assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]


@pytest.mark.parametrize("world,R,plt,store", [(2, 1, False, "auto"), (4, 2, False, "auto"), (2, 4, False, "auto"),
                                               (2, 2, True, "auto"), (4, 2, False, "packed"), (2, 4, False, "packed"),
                                               (4, 4, False, "auto")])
def test_multirank_layout_on_one_gpu(zd, oracle, ps, opk, world, R, plt, store):
    """the N>1 data path with the REAL kernels: `world` rank plans share this GPU, the all-to-all is
    emulated by chunk copies (chunk d of rank s's send buffer -> chunk s of rank d's receive buffer,
    i.e. all_to_all_single semantics); results must equal the single-process oracle."""
    import torch
    n = 64 if R < 4 else 128  # the folded z FFT needs N/R >= 32
    eig = oracle.synthetic_eigenmodes(32) if plt else None
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97) if plt else {}
    fmt = "RVdoubleZel"
    n = 256 if (world, R) == (4, 4) else n  # big enough for pruned tiles and compacted rows in every chunk
    plans = [zd.Plan(zd.make_params(n, icformat=fmt, stream_factor=R, store_mode=store, **kw), ps, eig=eig, rank=r,
                     nranks=world) for r in range(world)]
    nb = plans[0].exchange_bytes
    cb = nb // world
    Zq = plans[0].local_planes
    dt = zd.RECORD_DTYPES[fmt]
    rec = np.zeros((n, n * n), dtype=dt)
    assert plans[0].R == R
    for residue in range(plans[0].passes):  # R, or R/2 when the packed ZA store carries two residues per pass
        send = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(world)]
        recv = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(world)]
        for r in range(world):
            plans[r].stage_z(residue, send[r].data_ptr())
        torch.cuda.synchronize()
        for s in range(world):
            for d in range(world):
                recv[d][s * cb:(s + 1) * cb] = send[s][d * cb:(d + 1) * cb]
        for r in range(world):
            out = torch.zeros(Zq * n * n * dt.itemsize, dtype=torch.uint8, device="cuda")
            plans[r].stage_y(recv[r].data_ptr())
            plans[r].stage_x(residue, recv[r].data_ptr(), 0, Zq, out.data_ptr())
            torch.cuda.synchronize()
            host = out.cpu().numpy().view(dt).reshape(Zq, n * n)
            for i in range(Zq):
                rec[plans[r].plane_z(residue, i)] = host[i]
    stats = [p.stats() for p in plans]
    okw = dict(kw)
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat=fmt, **okw), opk, eig=eig,
                     eig_ppd=0 if eig is None else eig.shape[0])
    r = ref["records"].reshape(n, n * n)
    assert np.array_equal(rec["ijk"], r["ijk"])
    for f in ("d", "v"):
        assert _rel(rec[f], r[f]) < TOL
    var = sum(s["density_variance"] for s in stats)
    assert abs(var - ref["density_variance"]) <= TOL * ref["density_variance"]
    md = np.array([max((s["max_disp"][j] for s in stats), key=abs) for j in range(3)])
    assert _rel(md, ref["max_disp"]) < TOL
    for p in plans:
        p.close()


@pytest.mark.parametrize("plt", [False, True])
def test_packed_store_matches_reference_arrays(zd, oracle, ps, plt):
    """without ZD_qdensity the density field is not transformed (3 packed arrays; ZA: two residues per pass;
    density_variance from sum |D|^2): records and statistics equal those of the reference's 2 / 4 arrays"""
    n = 128
    eig = oracle.synthetic_eigenmodes(32) if plt else None
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97) if plt else {}
    p = zd.make_params(n, icformat="RVdoubleZel", stream_factor=4, **kw)
    plan = zd.Plan(p, ps, eig=eig)
    assert plan.narray == 3 and plan.plane_step == (1 if plt else 2) and plan.passes == (4 if plt else 2)
    plan.close()
    a = zd.generate(p, ps, eig=eig)
    p = zd.make_params(n, icformat="RVdoubleZel", stream_factor=4, store_mode="reference", **kw)
    plan = zd.Plan(p, ps, eig=eig)
    assert plan.narray == (4 if plt else 2) and plan.plane_step == 1 and plan.passes == 4
    plan.close()
    b = zd.generate(p, ps, eig=eig)
    if not plt:  # the round-1 packing (three arrays with Hermitian twins) as well
        c = zd.generate(zd.make_params(n, icformat="RVdoubleZel", stream_factor=4, store_mode="packed"), ps)
        assert _rel(c["records"]["d"], b["records"]["d"]) < 1e-13
    assert sorted(a["planes_seen"]) == list(range(n)) == sorted(b["planes_seen"])
    assert np.array_equal(a["records"]["ijk"], b["records"]["ijk"])
    for f in ("d", "v"):
        assert _rel(a["records"][f], b["records"][f]) < 1e-13
    assert abs(a["density_variance"] - b["density_variance"]) < 1e-12 * b["density_variance"]
    assert _rel(a["max_disp"], b["max_disp"]) < 1e-13


def test_bench_pipeline_driver(zd, oracle, ps, opk):
    """SlabPipeline + HipEngine (what bench.py times) delivers every plane once, matching the oracle"""
    import torch
    from zeldovich_plt_amd.parallel import HipEngine, SlabPipeline
    n, fmt = 64, "RVZel"
    plan = zd.Plan(zd.make_params(n, icformat=fmt, stream_factor=2), ps)
    pipe = SlabPipeline(HipEngine(plan, n), n, device="cuda", chunk_bytes=5 * n * n * 32)
    dt = zd.RECORD_DTYPES[fmt]
    rec = np.zeros((n, n * n), dtype=dt)
    seen = []

    def consume(zs, ring):
        torch.cuda.synchronize()
        host = ring.cpu().numpy()[:len(zs) * n * n * dt.itemsize].view(dt).reshape(len(zs), n * n)
        for i, z in enumerate(zs):
            rec[z] = host[i]
            seen.append(z)

    pipe.run(consume)
    assert sorted(seen) == list(range(n))
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat=fmt), opk)["records"].reshape(n, n * n)
    assert np.array_equal(rec["ijk"], ref["ijk"])
    assert _rel(rec["d"], ref["d"]) < 1e-6 and _rel(rec["v"], ref["v"]) < 1e-6
    plan.close()


@pytest.mark.parametrize("n,kw", [(256, dict()), (256, dict(k_cutoff=2.0)), (128, dict(k_cutoff=4.0, stream_factor=2)),
                                  (128, dict(corner_modes=1)), (256, dict(stream_factor=8))])
def test_pruned_columns_and_larger_grids(zd, oracle, ps, opk, n, kw):
    """sizes where whole z-FFT tiles are pruned (identically-zero (kx,ky) columns are neither generated,
    transformed nor read back), CornerModes (only the |k_i| == kmax rule prunes) and deep streaming"""
    _compare(zd, oracle, ps, opk, n, fmt="RVZel", **kw)


def test_plt_pruned_256(zd, oracle, ps, opk):
    eig = oracle.synthetic_eigenmodes(48)
    _compare(zd, oracle, ps, opk, 256, fmt="RVdoubleZel", eig=eig, qPLT=1, qPLTrescale=1, PLT_target_z=5.0,
             stream_factor=2)


def test_oversampling_invariance_on_gpu(zd, ps):
    """README: PPD=2N with k_cutoff=2 sub-sampled x2 equals PPD=N with k_cutoff=1 (size-independent property)"""
    lo = zd.generate(zd.make_params(128), ps)["records"]["d"]
    hi = zd.generate(zd.make_params(256, k_cutoff=2.0, stream_factor=2), ps)["records"]["d"]
    assert np.abs(hi[::2, ::2, ::2] - lo).max() < 1e-13 * np.abs(lo).max() + 1e-15


def test_stream_factor_invariance_at_1024(zd, ps):
    """full-size-ish property test (no oracle): the reductions of a PPD=1024 run do not depend on R"""
    a = zd.generate(zd.make_params(1024, icformat="RVZel", stream_factor=1), ps, collect=False)
    b = zd.generate(zd.make_params(1024, icformat="RVZel", stream_factor=4), ps, collect=False)
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-11 * a["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-11 * np.abs(a["max_disp"]).max()
    # the printed sanity check of the reference (zeldovich.cpp:987-996): rms pixel density vs sigma(R)
    rms = np.sqrt(a["density_variance"] / 1024.0 ** 3)
    pred = ps.sigmaR(720.0 / 1024 / 4.0) * 720.0 ** 1.5
    assert 0.5 < rms / pred < 1.5


def test_parseval_oversampling_property(zd, ps):
    """size-independent property used at full scale (PPD=8192 k_cutoff=2 vs PPD=4096):
    the same band-limited modes sampled on a 2x finer lattice give exactly 8x the sum of dens^2"""
    a = zd.generate(zd.make_params(128, icformat="RVZel"), ps, collect=False)
    b = zd.generate(zd.make_params(256, k_cutoff=2.0, icformat="RVZel", stream_factor=4), ps, collect=False)
    assert abs(b["density_variance"] / a["density_variance"] - 8.0) < 1e-12


@pytest.mark.skipif(not os.environ.get("ZD_RUN_SLOW"), reason="16 s; superseded by test_oversampled_planes_exact_at_full_size[2048] (records of "
                    "three planes, exact) and the direct sums of test_gpu_direct_sum.py: run with ZD_RUN_SLOW=1")
def test_full_size_properties_ppd4096_vs_2048(zd, ps):
    """BASELINE sizes through size-independent properties (no oracle run is feasible at 6.9e10 particles):
    PPD=4096 with k_cutoff=2 is phase-matched to PPD=2048 (README :52-54 of the reference) -> exactly 8x the sum of
    dens^2; the packed store (two residues per pass, Parseval) and the reference's two arrays agree at PPD=2048;
    the rms pixel density matches sigma(R) like the reference's own printed check (zeldovich.cpp:987-996)"""
    a = zd.generate(zd.make_params(2048, icformat="RVZel"), ps, collect=False)
    b = zd.generate(zd.make_params(4096, k_cutoff=2.0, icformat="RVZel"), ps, collect=False)
    assert abs(b["density_variance"] / a["density_variance"] - 8.0) < 1e-11
    c = zd.generate(zd.make_params(2048, icformat="RVZel", store_mode="reference"), ps, collect=False)
    assert abs(a["density_variance"] - c["density_variance"]) <= 1e-11 * c["density_variance"]
    assert np.abs(a["max_disp"] - c["max_disp"]).max() <= 1e-11 * np.abs(c["max_disp"]).max()
    rms = np.sqrt(a["density_variance"] / 2048.0 ** 3)
    pred = ps.sigmaR(720.0 / 2048 / 4.0) * 720.0 ** 1.5
    assert 0.5 < rms / pred < 1.5


def test_large_plt_packed_vs_reference_arrays(zd, oracle, ps):
    """PPD=1024 PLT+rescale (interpolated eigenmodes, two passes): the packed 3-array store and the reference's 4
    arrays give the same reductions"""
    eig = oracle.synthetic_eigenmodes(64)
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0, icformat="RVZel", stream_factor=2)
    a = zd.generate(zd.make_params(1024, **kw), ps, eig=eig, collect=False)
    b = zd.generate(zd.make_params(1024, store_mode="reference", **kw), ps, eig=eig, collect=False)
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-11 * b["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-11 * np.abs(b["max_disp"]).max()


@pytest.mark.parametrize("n,kw", [(64, dict()), (64, dict(stream_factor=2)), (128, dict(k_cutoff=2.0))])
def test_fnl_end_to_end(zd, oracle, ps, wmap_path, n, kw):
    """local primordial non-Gaussianity (next-row f.2): phi = D/M -> phi + f_NL phi^2 -> D = phi M, then the
    normal displacement path; the zero rule is bypassed in the second pass exactly as in the reference"""
    import ctypes as C
    fnl, ns, om = 2.0e4, 0.96, 0.31
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    p = zd.make_params(n, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **kw)
    got = zd.generate(p, ps)
    okw = {k: v for k, v in kw.items() if k != "stream_factor"}
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **okw), opk)
    lin = oracle.run(oracle.make_params(n, numblock=2, icformat="RVdoubleZel", **okw), opk)
    for f in ("d", "v"):
        for c in range(3):
            assert _rel(got["records"][f][..., c], ref["records"][f][..., c]) < TOL
    assert _rel(ref["records"]["d"], lin["records"]["d"]) > 1e-5  # the non-Gaussian term is really there
    assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]


@pytest.mark.parametrize("n,kw", [(128, dict(k_cutoff=2.0)), (128, dict(k_cutoff=2.0, ngpu=2)), (64, dict(k_cutoff=2.0, stream_factor=2)),
                                  (128, dict()), (128, dict(k_cutoff=2.0, f_NL=0.0)), (64, dict(f_NL=0.0, qPLT=1)),
                                  (192, dict(k_cutoff=2.0, f_NL=0.0, qdensity=1, stream_factor=4)),   # composite grid, six-field store: the density ring
                                  (160, dict(k_cutoff=4.0, f_NL=0.0, qdensity=1, stream_factor=2)),
                                  (100, dict(k_cutoff=2.0))])   # f_NL on the convolution kernels (launch_any_phi_nl / launch_any_phik)
def test_poisoned_buffers(zd, oracle, ps, wmap_path, n, kw):
    """with ZD_k_cutoff = 2 half of the phi store's column tiles are dead under the zero rule; the z stage
    never writes them and the y stage of the phi round must still deliver zeros there, because k_xphi / k_yfwd / k_zfwd read
    every column (only the x kernels of the main pass honour xdead_lo / hi).  Run in the -DZD_TESTING library with every store /
    ring / phi allocation pre-filled with NaN bytes (zd_test_poison): any kernel that reads what no kernel wrote turns the
    records into NaN.  Also the main pass (f_NL = 0; field and PLT stores)."""
    import ctypes as C
    kw = dict(kw)
    fnl, ns, om = kw.pop("f_NL", 2.0e4), 0.96, 0.31
    eig = oracle.synthetic_eigenmodes(32) if kw.get("qPLT") else None
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    T = zd.load_testing_library()
    T.zd_test_poison(1)
    try:
        got = zd.generate(zd.make_params(n, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **kw), ps, eig=eig, testing=True)
    finally:
        T.zd_test_poison(0)
    okw = {k: v for k, v in kw.items() if k not in ("stream_factor", "ngpu")}
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **okw), opk,
                     eig=eig, eig_ppd=0 if eig is None else eig.shape[0], want_density=bool(kw.get("qdensity")))
    for f in ("d", "v"):
        assert np.isfinite(got["records"][f]).all(), f
        for c in range(3):
            assert _rel(got["records"][f][..., c], ref["records"][f][..., c]) < TOL, (f, c)
    assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]
    if kw.get("qdensity"):  # round 4: the y stage skipped the dead column tiles of the density ring too, and k_xdens_q reads every column
        assert np.isfinite(got["density"]).all()
        assert _rel(got["density"], ref["density"]) < 1e-6


@pytest.mark.parametrize("ngpu,n,kw", [(2, 64, dict()), (4, 128, dict(stream_factor=2, exchange_planes=3)), (2, 64, dict(plt=True))])
def test_fnl_on_several_ranks(zd, oracle, ps, wmap_path, ngpu, n, kw):
    """ZD_f_NL with ZD_NumGPU > 1: the phi round travels to the XY ranks in plane groups, is transformed there (inverse y, x with
    phi + f_NL phi^2, forward y) and travels back; every rank then transforms its own rows along z (zd_multi.cpp phi_round).
    Ranks share this GPU (local transport)."""
    import ctypes as C
    kw = dict(kw)
    fnl, ns, om = 2.0e4, 0.96, 0.31
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(32)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    got = zd.generate(zd.make_params(n, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, ngpu=ngpu, **kw), ps, eig=eig)
    okw = {k: v for k, v in kw.items() if k not in ("stream_factor", "exchange_planes")}
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **okw), opk,
                     eig=eig, eig_ppd=0 if eig is None else eig.shape[0])
    for f in ("d", "v"):
        for c in range(3):
            assert _rel(got["records"][f][..., c], ref["records"][f][..., c]) < TOL, (f, c)
    assert sorted(got["planes_seen"]) == list(range(n))
    assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]


@pytest.mark.parametrize("n,kw", [(50, dict()), (96, dict(stream_factor=2)), (70, dict(plt=True, stream_factor=5))])
def test_fnl_on_any_even_ppd(zd, oracle, ps, wmap_path, n, kw):
    """ZD_f_NL on the convolution-transform PPDs (also 2^a 3^b ones: the composite kernels have no phi round): the forward
    transform of the real phi + f_NL phi^2 is the conjugate of its inverse transform"""
    import ctypes as C
    kw = dict(kw)
    fnl, ns, om = 2.0e4, 0.96, 0.31
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(24)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    got = zd.generate(zd.make_params(n, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **kw), ps, eig=eig)
    okw = {k: v for k, v in kw.items() if k != "stream_factor"}
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, **okw), opk,
                     eig=eig, eig_ppd=0 if eig is None else eig.shape[0])
    for f in ("d", "v"):
        for c in range(3):
            assert _rel(got["records"][f][..., c], ref["records"][f][..., c]) < TOL, (f, c)
    assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]


@pytest.mark.parametrize("n,kw", [
    (128, dict(k_cutoff=2.0)), (128, dict(k_cutoff=4.0, stream_factor=4)), (128, dict(k_cutoff=2.0, plt=True)),
    (128, dict(k_cutoff=2.0, store_mode="packed")), (128, dict(k_cutoff=2.0, store_mode="packed", plt=True)),
    (128, dict(k_cutoff=2.0, qdensity=1)), (128, dict(k_cutoff=2.0, qdensity=2)), (128, dict(k_cutoff=2.0, store_mode="reference", plt=True)),
    (128, dict(k_cutoff=2.0, ngpu=2, exchange_planes=3)), (128, dict(k_cutoff=2.0, ngpu=2, plt=True)),
    (96, dict(k_cutoff=2.0, stream_factor=2)), (160, dict(k_cutoff=2.0, stream_factor=2, plt=True)), (224, dict(k_cutoff=4.0, stream_factor=2)),
    (192, dict(k_cutoff=2.0, stream_factor=2, ngpu=2)), (192, dict(k_cutoff=2.0, stream_factor=2, ngpu=2, qdensity=1)),
    (192, dict(k_cutoff=2.0, stream_factor=2, ngpu=2, plt=True)), (256, dict(k_cutoff=4.0, ngpu=4, exchange_planes=2)),
    (100, dict(k_cutoff=2.0)), (100, dict(k_cutoff=2.0, qdensity=1)), (100, dict(k_cutoff=2.0, plt=True)),
    (64, dict(k_cutoff=2.0, version=1, numblock=4)),
])
def test_poison_sweep_of_the_stores(zd, oracle, ps, n, kw):
    """Every store / packing / transform family with ZD_k_cutoff > 1 (pruned columns: the z stage leaves the dead column tiles of the
    stores unwritten) in the -DZD_TESTING library with all stores, rings and phi fields starting as NaN bytes (zd_test_poison): the
    records and density planes must be finite and equal to the product library's ordinary run.  (Round 4 found the density ring of
    the six-field store this way; ADVICE r3 the phi store.)"""
    kw = dict(kw)
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(32)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    p = zd.make_params(n, icformat="RVdoubleZel", **kw)
    base = zd.generate(p, ps, eig=eig)
    T = zd.load_testing_library()
    T.zd_test_poison(1)
    try:
        got = zd.generate(p, ps, eig=eig, testing=True)
    finally:
        T.zd_test_poison(0)
    if base["records"] is not None:
        for f in ("d", "v"):
            assert np.isfinite(got["records"][f]).all(), f
            assert _rel(got["records"][f], base["records"][f]) < 1e-13, f
    if kw.get("qdensity"):
        assert np.isfinite(got["density"]).all()
        assert _rel(got["density"], base["density"]) < 1e-6
    assert abs(got["density_variance"] - base["density_variance"]) <= 1e-12 * base["density_variance"]


def test_fnl_several_ranks_equal_one_rank_at_512(zd, ps):
    """ZD_f_NL at PPD = 512: four ranks (several plane groups both ways) against the single-GPU path, records of sample planes"""
    n, zs = 512, (3, 259, 510)
    got = {}
    for ngpu in (1, 4):
        planes = {}
        p = zd.make_params(n, icformat="RVZel", f_NL=2.0e4, n_s=0.96, Omega_M=0.31, ngpu=ngpu, exchange_planes=7 if ngpu > 1 else 0)
        out = zd.generate_planes(p, ps, lambda z, rec: planes.__setitem__(z, rec.copy()) if z in zs else None)
        got[ngpu] = (planes, out)
    for z in zs:
        a, b = got[1][0][z], got[4][0][z]
        assert np.abs(a["d"] - b["d"]).max() <= 1e-6 * np.abs(a["d"]).max(), z
    assert abs(got[1][1]["density_variance"] - got[4][1]["density_variance"]) <= 1e-11 * got[1][1]["density_variance"]


def test_fnl_through_the_staged_api(zd, oracle, ps, wmap_path):
    """ZD_f_NL through zd_plan_create + the staged pipeline (what bench.py drives): the phi round runs at plan creation, the
    plan owns PhiK; every plane against the oracle"""
    import ctypes as C
    import torch
    from zeldovich_plt_amd.parallel import HipEngine, SlabPipeline
    n, fmt = 64, "RVdoubleZel"
    fnl, ns, om = 2.0e4, 0.96, 0.31
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    plan = zd.Plan(zd.make_params(n, icformat=fmt, stream_factor=2, f_NL=fnl, n_s=ns, Omega_M=om), ps)
    pipe = SlabPipeline(HipEngine(plan, n), n, device="cuda", chunk_bytes=7 * n * n * 56)
    dt = zd.RECORD_DTYPES[fmt]
    rec = np.zeros((n, n * n), dtype=dt)
    seen = []

    def consume(zs, ring):
        torch.cuda.synchronize()
        host = ring.cpu().numpy()[:len(zs) * n * n * dt.itemsize].view(dt).reshape(len(zs), n * n)
        for i, z in enumerate(zs):
            rec[z] = host[i]
            seen.append(z)

    pipe.run(consume)
    assert sorted(seen) == list(range(n))
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat=fmt, f_NL=fnl, n_s=ns, Omega_M=om), opk)["records"].reshape(n, n * n)
    assert _rel(rec["d"], ref["d"]) < TOL and _rel(rec["v"], ref["v"]) < TOL
    plan.close()
    with pytest.raises(RuntimeError):  # the phi round needs every plane on one rank
        zd.Plan(zd.make_params(n, icformat=fmt, f_NL=fnl, n_s=ns, Omega_M=om), ps, rank=0, nranks=2)


def test_fnl_with_plt(zd, oracle, ps, wmap_path):
    import ctypes as C
    eig = oracle.synthetic_eigenmodes(32)
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), 1.0)
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_NL=1.0e4)
    got = zd.generate(zd.make_params(64, icformat="RVdoubleZel", **kw), ps, eig=eig)
    ref = oracle.run(oracle.make_params(64, numblock=2, icformat="RVdoubleZel", **kw), opk, eig=eig, eig_ppd=32)
    for f in ("d", "v"):
        assert _rel(got["records"][f], ref["records"][f]) < TOL


@pytest.mark.parametrize("ngpu,n,kw", [
    (2, 128, dict(stream_factor=2)),                                   # field store, two ranks, one exchange group
    (2, 128, dict(stream_factor=2, exchange_planes=5)),                # 32 planes per rank in groups of 5 (last: 2)
    (4, 256, dict(stream_factor=4, k_cutoff=2.0, fmt="RVZel", exchange_planes=3)),  # compacted rows, pruned tiles
    (2, 128, dict(store_mode="reference", exchange_planes=7)),
    (2, 64, dict(store_mode="reference")),                             # the reference's arrays with Hermitian twins
    (4, 128, dict(stream_factor=2, store_mode="packed")),
    (2, 128, dict(stream_factor=2, plt=True)),
    (2, 192, dict(stream_factor=2)),                                   # PPD = 2^6 3: 48 planes per rank (composite transforms)
    (4, 192, dict(stream_factor=4, k_cutoff=2.0, exchange_planes=5)),  # 12 planes per rank in groups of 5
    pytest.param(2, 288, dict(stream_factor=2, exchange_planes=7), marks=pytest.mark.slow),  # 2^5 3^2 (oracle-bound: its plain DFT is O(N^4); the 192 rows stay)
    (2, 128, dict(qdensity=1, fmt="RVZel", exchange_planes=9)),        # density plane beside the records (reference arrays)
    (4, 128, dict(qdensity=2, stream_factor=2)),                       # density only
    (2, 64, dict(qoneslab=17)),                                        # one slab: finished by the single-GPU path
])
def test_native_multi_gpu_driver(zd, oracle, ps, opk, ngpu, n, kw):
    """ZD_NumGPU > 1 through zd_generate: one host thread per rank, exchange in plane groups + pipelined XY stages inside
    the library (csrc/zd_multi.cpp).  This box has one GPU, so the ranks share it and pull their slices with device
    copies (the `local` transport; RCCL needs distinct GPUs) — kernels, layouts, ring and plane-group pipeline are the real
    ones.  Result == single-process oracle."""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(32)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    if "qoneslab" in kw:
        p = zd.make_params(n, ngpu=ngpu, **kw)
        got = zd.generate(p, ps)
        ref = oracle.run(oracle.make_params(n, numblock=2, qoneslab=kw["qoneslab"]), opk)
        z = kw["qoneslab"]
        assert got["planes_seen"] == [z]
        assert _rel(got["records"]["d"][z], ref["records"]["d"][z]) < TOL
        assert abs(got["density_variance"] - ref["density_variance"]) <= TOL * ref["density_variance"]
        return
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, ngpu=ngpu, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("ngpu,n,kw", [
    (2, 128, dict(stream_factor=2)),                                   # field store, one exchange group
    (4, 128, dict(stream_factor=2, exchange_planes=3)),                # 16 planes per rank in groups of 3 (last: 1)
    (2, 128, dict(store_mode="reference", exchange_planes=7, qdensity=1, fmt="RVZel")),
    (4, 256, dict(stream_factor=4, k_cutoff=2.0, exchange_planes=5, plt=True)),
    (8, 128, dict(stream_factor=2, exchange_planes=1)),                # eight ranks
    (2, 192, dict(stream_factor=2, exchange_planes=5)),                # composite grid
])
def test_rccl_branch_on_the_loopback_emulation(zd, oracle, ps, opk, ngpu, n, kw):
    """the RCCL branch of the plane-group exchange (grouped ncclSend / ncclRecv per group on the communication stream, event
    ordering against the XY stages, ring-slot reuse) with the ranks as threads on this one GPU, on an in-process emulation of
    the RCCL calls (zd_test_generate_loopback): real RCCL refuses two ranks on one device.  Result == oracle."""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(32)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, ngpu=ngpu, loopback=True, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


def test_rccl_branch_fnl_round_trip_on_the_loopback_emulation(zd, oracle, ps, wmap_path):
    """the phi round's forward and REVERSE plane-group exchange (zd_multi.cpp phi_round, RCCL branch) on the loopback emulation"""
    import ctypes as C
    n, fnl, ns, om = 128, 2.0e4, 0.96, 0.31
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    got = zd.generate(zd.make_params(n, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om, ngpu=4, stream_factor=2, exchange_planes=3),
                      ps, loopback=True)
    ref = oracle.run(oracle.make_params(n, numblock=2, icformat="RVdoubleZel", f_NL=fnl, n_s=ns, Omega_M=om), opk)
    for f in ("d", "v"):
        assert _rel(got["records"][f], ref["records"][f]) < TOL
    assert sorted(got["planes_seen"]) == list(range(n))


def test_rccl_communicator_single_rank(zd, ps):
    """the library's RCCL binding (lazy dlopen, ncclGetUniqueId, ncclCommInitRank) on the one GPU of this box; a pass
    through zd_plan_run_pass with that communicator"""
    import torch
    comm = zd.Comm(0, 1, lambda raw: raw)
    n = 64
    plan = zd.Plan(zd.make_params(n, icformat="RVZel", stream_factor=2), ps)
    store = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
    rec = torch.empty(plan.local_planes * n * n * 32, dtype=torch.uint8, device="cuda")
    seen = []
    plan.run_pass(0, store.data_ptr(), rec.data_ptr(), plan.local_planes, comm=comm,
                  consume=lambda first, cnt, ptr, st: seen.append((first, cnt)))
    torch.cuda.synchronize()
    assert seen == [(0, plan.local_planes)]
    plan.close()
    comm.close()


def test_randomised_option_sweep_vs_oracle(zd, oracle, wmap_path):
    """a seeded sweep over option combinations (size, record format, stream factor, store mode, ZD_k_cutoff, CornerModes,
    fixed amplitudes, smoothing, PLT / rescale / f_cluster, seed, box) — every run against the oracle"""
    rng = np.random.default_rng(20261003)
    fmts = ["RVdoubleZel", "RVZel", "Zeldovich", "ZelSimple"]
    done = 0
    for trial in range(28 if os.environ.get("ZD_RUN_SLOW") else 20):  # (the same seeded sequence; ZD_RUN_SLOW=1 runs the last 8 too)
        n = int(rng.choice([64, 64, 128, 128, 256]))
        plt = bool(rng.integers(0, 2)) and n <= 128
        kw = dict(seed=int(rng.integers(1, 2 ** 31 - 1)), k_cutoff=float(rng.choice([1.0, 1.0, 1.5, 2.0, 4.0])),
                  corner_modes=int(rng.integers(0, 2)), f_cluster=float(rng.choice([1.0, 0.93])),
                  boxsize=float(rng.choice([720.0, 720.0, 90.0, 2000.0])))
        R = int(rng.choice([1, 2, 4]))
        if n // R < 32:
            R = 1
        kw["stream_factor"] = R
        kw["store_mode"] = str(rng.choice(["auto", "auto", "reference", "packed"]))
        if rng.integers(0, 4) == 0:
            kw["qdensity"] = int(rng.choice([1, 2]))
        eig = None
        if plt and kw.get("qdensity", 0) != 2:
            eig = oracle.synthetic_eigenmodes(int(rng.choice([16, 24, 32, 48, 64])), seed=int(rng.integers(0, 100)))
            kw.update(qPLT=1, qPLTrescale=int(rng.integers(0, 2)), PLT_target_z=float(rng.choice([3.0, 9.0])))
        pkw = dict(fix_to_mean=int(rng.integers(0, 2)), Pk_smooth=float(rng.choice([0.0, 0.0, 0.5])))
        box = kw["boxsize"]
        ps = zd.PowerSpectrum.from_file(wmap_path, box, **pkw)
        opk = oracle.pk_from_file(wmap_path, box, **pkw)
        fmt = str(rng.choice(fmts))
        print("trial", trial, n, fmt, kw, pkw, "eig" if eig is not None else "")
        _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, **kw)
        done += 1
    assert done == (28 if os.environ.get("ZD_RUN_SLOW") else 20)


def test_two_ranks_equal_one_rank_at_1024(zd, ps):
    """ZD_NumGPU = 2 (ranks sharing this GPU, several exchange groups) against the single-rank run at PPD = 1024: the same
    records on sample planes, the same reductions"""
    n = 1024
    got = {}
    for ngpu in (1, 2):
        planes = {}

        def keep(z, rec):
            if z in (3, 517, 1022):
                planes[z] = rec.copy()

        p = zd.make_params(n, icformat="RVZel", ngpu=ngpu, exchange_planes=100 if ngpu > 1 else 0)
        out = zd.generate_planes(p, ps, keep)
        got[ngpu] = (planes, out)
    for z in (3, 517, 1022):
        a, b = got[1][0][z], got[2][0][z]
        assert np.array_equal(a["ijk"], b["ijk"])
        assert np.abs(a["d"] - b["d"]).max() <= 1e-6 * np.abs(a["d"]).max()
    a, b = got[1][1], got[2][1]
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-11 * a["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-11 * np.abs(a["max_disp"]).max()
    assert a["planes"] == b["planes"] == n


def test_eight_ranks_equal_one_rank(zd, ps):
    """the geometry the 8-GPU scaling run uses (8 ranks: rows ky = rank mod 8, Zq = L/8 planes per rank, plane-group exchange),
    with the ranks as threads sharing this one GPU (local transport): PPD = 1024 — records of sample planes equal to the
    single-rank run; PPD = 4096 (BASELINE C4's size; R = 32 so that eight stores fit one GPU) — the reductions over every
    particle (max_disp per component, sum delta^2) equal to the single-rank run"""
    n, zs = 1024, (5, 517, 1022)  # (PPD = 2048 until round 4: 200 GB of records over PCIe for three sample planes)
    got = {}
    for ngpu in (1, 8):
        planes = {}
        p = zd.make_params(n, icformat="ZelSimple", ngpu=ngpu, stream_factor=8 if ngpu > 1 else 0, exchange_planes=5 if ngpu > 1 else 0)
        out = zd.generate_planes(p, ps, lambda z, rec: planes.__setitem__(z, rec["d"].copy()) if z in zs else None)
        got[ngpu] = (planes, out)
    for z in zs:
        a, b = got[1][0][z], got[8][0][z]
        assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max(), z
    assert got[1][1]["planes"] == got[8][1]["planes"] == n
    if not os.environ.get("ZD_RUN_SLOW"):  # (the PPD = 4096 reductions: 7 s; the geometry is the same code at every size)
        return
    a = zd.generate(zd.make_params(4096, icformat="RVZel"), ps, collect=False)
    b = zd.generate(zd.make_params(4096, icformat="RVZel", ngpu=8, stream_factor=32, exchange_planes=2), ps, collect=False)
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-11 * a["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-11 * np.abs(a["max_disp"]).max()


@pytest.mark.parametrize("n,kw", [
    (96, dict(stream_factor=2)),                    # 32 * 3
    (160, dict(stream_factor=2, fmt="RVZel")),      # 32 * 5
    pytest.param(288, dict(stream_factor=6), marks=pytest.mark.slow),  # 32 * 9, three passes (oracle-bound; Q = 9 density: 2304 in the one-mode test)
    (192, dict(stream_factor=4, k_cutoff=2.0)),     # pruned columns
    (192, dict(stream_factor=2, ngpu=2)),           # two ranks that exchange the six fields
])
def test_density_on_composite_grids_vs_oracle(zd, oracle, ps, opk, n, kw):
    """ZD_qdensity = 1 on the composite grids (round 4): the ZA field store carries the density sums D of the two
    residues as fields 4, 5 (generator kind GENF_ZAFD), the y stage builds one more array delta_r0 + i delta_r1 and `k_xdens_q`
    writes the float32 density planes (src/output.cpp:93-101,217-224) — instead of the ~6x slower convolution path.  Records,
    density planes, max_disp and density_variance against the oracle."""
    kw = dict(kw)
    plan = zd.Plan(zd.make_params(n, qdensity=1, **{k: v for k, v in kw.items() if k not in ("fmt", "ngpu")}), ps)
    assert plan.store_mode == "fields" and plan.plane_step == 2  # the six-field store, not the reference arrays of the convolution path
    plan.close()
    got, ref = _compare(zd, oracle, ps, opk, n, qdensity=1, **kw)
    assert got["density"] is not None and np.abs(ref["density"]).max() > 0
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("n,kw", [
    (96, dict(stream_factor=2)),                      # 32 * 3
    (160, dict(stream_factor=2, k_cutoff=2.0)),       # 32 * 5, pruned columns
    (224, dict(stream_factor=2)),                     # 32 * 7
    (192, dict(stream_factor=2, ngpu=2)),             # two ranks that exchange the six fields
])
def test_density_only_on_composite_grids_vs_oracle(zd, oracle, ps, opk, n, kw):
    """ZD_qdensity = 2 — density only: no displacements, no records (src/zeldovich.cpp:303,872-876, src/output.cpp:94,207-224) — on the
    composite kernels (round 5; before: the ~6x slower convolution path): the six-field ZA store, of which only the
    density array is built and transformed.  Density planes (float32) and density_variance against the oracle; no records, and
    max_disp stays zero as in the reference (WriteParticlesSlab never touches it)."""
    kw = dict(kw)
    plan = zd.Plan(zd.make_params(n, qdensity=2, **{k: v for k, v in kw.items() if k != "ngpu"}), ps)
    assert plan.store_mode == "fields"  # not the single reference array of the convolution path
    plan.close()
    got, ref = _compare(zd, oracle, ps, opk, n, qdensity=2, **kw)
    assert got["records"] is None and ref["records"] is None
    assert got["density"] is not None and np.abs(ref["density"]).max() > 0
    assert np.abs(got["max_disp"]).max() == 0.0
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("n,kw", [
    (96, dict(stream_factor=1)),                                   # 32 * 3: z lines of 96, the density half at 2 x 48
    (96, dict(stream_factor=2)),                                   # z lines of 48, the density half at 4 x 24 (= 8 * 3)
    pytest.param(160, dict(stream_factor=1, qPLTrescale=1), marks=pytest.mark.slow),  # 32 * 5
    pytest.param(224, dict(stream_factor=1), marks=pytest.mark.slow),                 # 32 * 7
    (192, dict(stream_factor=2, qPLTrescale=1)),                   # two passes, each with its own density-only pass (R = 4) in front
    (192, dict()),                                                 # stream factor left to the library
    (192, dict(stream_factor=2, ngpu=2, pass_groups=2)),           # two GPUs, one pass each: every rank composes its own pass
])
def test_plt_with_density_on_composite_grids_vs_oracle(zd, oracle, ps, opk, n, kw):
    """ZD_qPLT with ZD_qdensity = 1 on the composite kernels (round 5; before: the ~6x slower convolution path): the PLT
    field store has no density field, so every pass is preceded by a density-only pass of the six-field ZA store at twice the
    stream factor — the same planes — on the same store (`plan_create_ex`, `plt_dens_split`).  Records, density planes, density_variance and
    max_disp against the oracle."""
    eig = oracle.synthetic_eigenmodes(32)
    kw = dict(kw)
    if kw.get("qPLTrescale"):
        kw.update(PLT_target_z=5.0, z_initial=49.0)
    plan = zd.Plan(zd.make_params(n, qPLT=1, qdensity=1, **{k: v for k, v in kw.items() if k not in ("ngpu", "pass_groups")}), ps, eig=eig)
    if kw.get("stream_factor"):
        assert plan.store_mode == "fields"  # the composite PLT field store, not the convolution path's reference arrays
    plan.close()
    got, ref = _compare(zd, oracle, ps, opk, n, eig=eig, qPLT=1, qdensity=1, **kw)
    assert got["records"] is not None and got["density"] is not None and np.abs(ref["density"]).max() > 0
    assert sorted(got["planes_seen"]) == list(range(n))


def test_density_only_ignores_plt_on_composite_grids(zd, oracle, ps, opk):
    """ZD_qdensity = 2 with ZD_qPLT set: the displacement arrays are never built (src/zeldovich.cpp:303,440), so the run is the ZA
    density-only run — on the composite kernels too (it used to fall to the convolution path because of the PLT flag)"""
    eig = oracle.synthetic_eigenmodes(32)
    plan = zd.Plan(zd.make_params(160, qPLT=1, qdensity=2, stream_factor=2), ps, eig=eig)
    assert plan.store_mode == "fields"
    plan.close()
    got, ref = _compare(zd, oracle, ps, opk, 160, eig=eig, qPLT=1, qdensity=2, stream_factor=2)
    assert got["records"] is None and got["density"] is not None


def test_density_on_composite_grid_short_z_lines_vs_convolution_path(zd, ps):
    """PPD = 480 = 32 * 15 with ZD_qdensity = 1 at R = 8: z lines of 60 = 4 * 15, i.e. the generator's short walk (four z rows per
    thread, `k_genf<4, GENF_ZAFD>`) and the 4-element z transform with six fields — beyond the oracle's O(N^4) plain DFT, so the
    comparator is the library's OTHER implementation of the same option: the reference arrays through the convolution kernels
    (`store_mode = reference`), records and density planes"""
    n = 480
    a = zd.generate(zd.make_params(n, icformat="RVZel", qdensity=1, stream_factor=8), ps)
    b = zd.generate(zd.make_params(n, icformat="RVZel", qdensity=1, store_mode="reference"), ps)
    assert a["stream_factor"] == 8
    for f in ("d", "v"):
        scale = np.abs(b["records"][f]).max()
        assert scale > 0 and np.abs(a["records"][f] - b["records"][f]).max() <= 2e-6 * scale, f  # float32 records
    assert np.array_equal(a["records"]["ijk"], b["records"]["ijk"])
    ds = np.abs(b["density"]).max()
    assert ds > 0 and np.abs(a["density"] - b["density"]).max() <= 2e-6 * ds
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-10 * b["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-10 * np.abs(b["max_disp"]).max()


@pytest.mark.parametrize("n", [24, 72, 216, 48, 144, 432, 96, 288, 864, 192, 576, 1728, 384, 1152, 3456, 768, 2304, 6912,
                               1536, 4608, 3072])
@pytest.mark.parametrize("kind", [0, 1])
def test_fft_lines_composite_lengths(zd, n, kind):
    """lengths 2^a 3^b (b <= 3): Q = 3^b decimated sub-lines through the power-of-two register engine + Q-point outer
    transforms (csrc/zd_fft_q.h), both LDS layouts, vs numpy"""
    rng = np.random.default_rng(n + kind)
    lines = 8
    x = rng.standard_normal((lines, n)) + 1j * rng.standard_normal((lines, n))
    got = zd.test_fft(x, kind)
    ref = np.fft.ifft(x, axis=1) * n
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-14, err


@pytest.mark.parametrize("n,kw", [
    (96, dict(stream_factor=2)),                               # 96 = 32 * 3, z lines of 48
    (192, dict(stream_factor=2, fmt="RVZel")),                  # 192 = 64 * 3
    (192, dict(stream_factor=4, k_cutoff=2.0)),
    (288, dict(stream_factor=2)),                              # 288 = 32 * 9
    (96, dict(stream_factor=2, fmt="Zeldovich", k_cutoff=1.5)),
    pytest.param(288, dict(stream_factor=6, ngpu=2, fmt="RVZel"), marks=pytest.mark.slow),  # a stream factor that is not a power of two (z lines of 48, 3 passes), on two ranks that exchange (24 planes per rank and pass)
    pytest.param(288, dict(stream_factor=2, k_cutoff=2.0, fmt="ZelSimple"), marks=pytest.mark.slow),  # (Q = 27 sizes start at 864: beyond the oracle's O(N^4) plain DFT;
                                                                  # covered by test_non_power_of_two_oversampling_invariance and test_non_power_of_two_short_z_lines)
])
def test_non_power_of_two_ppd_vs_oracle(zd, oracle, ps, opk, n, kw):
    """PPD = 2^a 3^b (the reference plans any length with FFTW, src/zeldovich.cpp:61-66): composite-length
    transforms on the ZA field store against the oracle (whose non-power-of-two path is a plain DFT)"""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    plan = zd.Plan(zd.make_params(n, icformat=fmt, **kw), ps)
    assert plan.store_mode == "fields" and plan.plane_step == 2
    plan.close()
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("n,kw", [
    (96, dict(stream_factor=2, ppd_e=32)),                       # exact-stride eigenmode lookup is impossible (32 does not divide 96): trilinear
    pytest.param(192, dict(stream_factor=4, ppd_e=64, resc=0), marks=pytest.mark.slow),
    (192, dict(stream_factor=2, ppd_e=24, fmt="RVZel")),
    (192, dict(stream_factor=2, ppd_e=32, ngpu=2)),              # two ranks
    (192, dict(stream_factor=2, ppd_e=32, version=1, numblock=4)),  # legacy streams on a composite grid
])
def test_non_power_of_two_ppd_plt_vs_oracle(zd, oracle, ps, opk, n, kw):
    """PLT (+ rescale) on PPD = 2^a 3^b — the production Abacus configuration (PPD = 6912 with ZD_qPLT): the PLT field store
    (six half-space sums) through the composite-transform kernels, against the oracle"""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = oracle.synthetic_eigenmodes(kw.pop("ppd_e"))
    resc = kw.pop("resc", 1)
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, qPLT=1, qPLTrescale=resc, PLT_target_z=5.0, f_cluster=0.97, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("n", [40, 200, 1000, 80, 160, 1280, 5120, 240, 960, 400, 3200, 720, 2400, 2000, 4000])
@pytest.mark.parametrize("kind", [0, 1])
def test_fft_lines_radix5_lengths(zd, n, kind):
    """lengths 2^a 3^b 5^c (round 3): Q = 5, 15, 25, 45, 75, 125 decimated sub-lines through the power-of-two register engine + a
    mixed radix-3 / radix-5 outer transform (csrc/zd_fft_q.h outer_stages), both LDS layouts, vs numpy"""
    rng = np.random.default_rng(n + kind)
    lines = 8
    x = rng.standard_normal((lines, n)) + 1j * rng.standard_normal((lines, n))
    got = zd.test_fft(x, kind)
    ref = np.fft.ifft(x, axis=1) * n
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-14, err


@pytest.mark.parametrize("n,kw", [
    (160, dict(stream_factor=2)),                                       # 160 = 32 * 5, z lines of 80 = 16 * 5
    (160, dict(stream_factor=2, k_cutoff=2.0, fmt="RVZel")),
    (160, dict(stream_factor=2, plt=32)),                                # PLT + rescale, interpolated 32^3 table
    pytest.param(320, dict(stream_factor=4, fmt="ZelSimple"), marks=pytest.mark.slow),  # 320 = 64 * 5, z lines of 80 (oracle-bound, 11 s; 160 stays, 320 <-> 160 ... in the oversampling links)
])
def test_radix5_ppd_vs_oracle(zd, oracle, ps, opk, n, kw):
    """PPD = 2^a 5 on the composite-transform kernels (field stores) against the oracle, whose non-power-of-two path is a plain DFT"""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = None
    if "plt" in kw:
        eig = oracle.synthetic_eigenmodes(kw.pop("plt"))
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    plan = zd.Plan(zd.make_params(n, icformat=fmt, **kw), ps, eig=eig)
    assert plan.store_mode == "fields"
    plan.close()
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("n", [56, 112, 448, 1792, 7168, 336, 2688, 560, 2240, 784, 6272, 2160, 4320, 8640])
@pytest.mark.parametrize("kind", [0, 1])
def test_fft_lines_radix7_lengths(zd, n, kind):
    """lengths 2^a * {7, 21, 35, 49} and 2^a * 135 (round 4; the reference plans any length with FFTW, src/zeldovich.cpp:61-66): radix-7
    stages in the outer transform of csrc/zd_fft_q.h, both LDS layouts, vs numpy"""
    rng = np.random.default_rng(n + kind)
    lines = 8
    x = rng.standard_normal((lines, n)) + 1j * rng.standard_normal((lines, n))
    got = zd.test_fft(x, kind)
    ref = np.fft.ifft(x, axis=1) * n
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-14, err


@pytest.mark.parametrize("n,kw", [
    (224, dict(stream_factor=2)),                                        # 224 = 32 * 7, z lines of 112 = 16 * 7
    (224, dict(stream_factor=2, plt=32, fmt="RVZel")),                    # PLT + rescale, interpolated 32^3 table
    (224, dict(stream_factor=2, qdensity=1, k_cutoff=2.0, fmt="ZelSimple")),  # six-field store: density on the composite kernels
])
def test_radix7_ppd_vs_oracle(zd, oracle, ps, opk, n, kw):
    """PPD = 2^a 7 on the composite-transform kernels (field stores) against the oracle, whose non-power-of-two path is a plain DFT.
    (The larger radix-7 grids hang off this one through the oversampling links of test_gpu_baseline_regime.py
    test_radix7_oversampled_planes.)"""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = None
    if "plt" in kw:
        eig = oracle.synthetic_eigenmodes(kw.pop("plt"))
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    plan = zd.Plan(zd.make_params(n, icformat=fmt, **kw), ps, eig=eig)
    assert plan.store_mode == "fields"
    plan.close()
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


def test_radix5_oversampling_and_stream_invariance(zd, ps):
    """5-smooth grids beyond the oracle's reach: 1600 (= 64 * 25, k_cutoff = 2) <-> 800 (= 32 * 25) and 960 (= 64 * 15) <-> 480 through
    the oversampling invariant on sample planes; R-invariance of the reductions at 1280 (= 256 * 5).  (4000 = 32 * 125 against 2000 on
    the convolution kernels: test_oversampled_planes_exact_at_full_size.)"""
    for n, Rlo, Rhi in ((800, 2, 4), (480, 2, 4)):
        zs = (1, n // 2 + 1, n - 1)
        lo, hi = {}, {}
        zd.generate_planes(zd.make_params(n, icformat="Zeldovich", stream_factor=Rlo), ps,
                           lambda z, rec: lo.__setitem__(z, rec["d"].copy()) if z in zs else None)
        zd.generate_planes(zd.make_params(2 * n, icformat="Zeldovich", k_cutoff=2.0, stream_factor=Rhi), ps,
                           lambda z, rec: hi.__setitem__(z // 2, rec["d"][::2, ::2].copy()) if (z % 2 == 0 and z // 2 in zs) else None)
        assert sorted(lo) == sorted(hi) == sorted(zs)
        for z in zs:
            assert np.abs(hi[z] - lo[z]).max() < 1e-12 * np.abs(lo[z]).max() + 1e-15, (n, z)
    a = zd.generate(zd.make_params(1280, icformat="RVZel", stream_factor=2), ps, collect=False)
    b = zd.generate(zd.make_params(1280, icformat="RVZel", stream_factor=8), ps, collect=False)
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-11 * a["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-11 * np.abs(a["max_disp"]).max()


def test_non_power_of_two_oversampling_invariance(zd, ps):
    """PPD = 2N with k_cutoff = 2 at even sites == PPD = N (README), across the composite sizes: 192 <-> 96, 576 <-> 288,
    1728 <-> 864 (27 * 64 / 27 * 32), on sample planes (a full PPD = 1728 record array would be 290 GB of host memory);
    R-invariance of the reductions at 1152"""
    for n in (96, 288, 864):
        zs = (1, n // 2 + 1, n - 1)
        lo, hi = {}, {}
        zd.generate_planes(zd.make_params(n, icformat="Zeldovich", stream_factor=2), ps,
                           lambda z, rec: lo.__setitem__(z, rec["d"].copy()) if z in zs else None)
        zd.generate_planes(zd.make_params(2 * n, icformat="Zeldovich", k_cutoff=2.0, stream_factor=4), ps,
                           lambda z, rec: hi.__setitem__(z // 2, rec["d"][::2, ::2].copy()) if (z % 2 == 0 and z // 2 in zs) else None)
        assert sorted(lo) == sorted(hi) == sorted(zs)
        for z in zs:
            assert np.abs(hi[z] - lo[z]).max() < 1e-12 * np.abs(lo[z]).max() + 1e-15, (n, z)
    a = zd.generate(zd.make_params(1152, icformat="RVZel", stream_factor=2), ps, collect=False)
    b = zd.generate(zd.make_params(1152, icformat="RVZel", stream_factor=8), ps, collect=False)
    assert abs(a["density_variance"] - b["density_variance"]) <= 1e-11 * a["density_variance"]
    assert np.abs(a["max_disp"] - b["max_disp"]).max() <= 1e-11 * np.abs(a["max_disp"]).max()


def test_non_power_of_two_short_z_lines(zd, oracle, ps):
    """z lines of 108 = 4 * 27 (generator threads walk 4 z rows instead of 16; one thread per 4-point sub-line in the z
    transform) — what lets PPD = 6912 run on ONE GPU at ZD_k_cutoff = 1 (R = 64).  PPD = 864 at R = 8 against R = 2 (z lines
    of 432), ZA and PLT, on sample planes; 1728 (k_cutoff = 2, R = 16) against 864"""
    n = 864
    zs = (2, n // 2 + 5, n - 3)
    eig = oracle.synthetic_eigenmodes(32)
    for kw, e in ((dict(icformat="Zeldovich"), None),
                  (dict(icformat="RVdoubleZel", qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97), eig)):
        a, b = {}, {}
        zd.generate_planes(zd.make_params(n, stream_factor=2, **kw), ps, lambda z, rec: a.__setitem__(z, rec.copy()) if z in zs else None, eig=e)
        zd.generate_planes(zd.make_params(n, stream_factor=8, **kw), ps, lambda z, rec: b.__setitem__(z, rec.copy()) if z in zs else None, eig=e)
        assert sorted(a) == sorted(b) == sorted(zs)
        for z in zs:
            for f in ("d", "v"):
                if f in a[z].dtype.names:
                    assert np.abs(a[z][f]).max() > 0
                    assert np.abs(a[z][f] - b[z][f]).max() <= 1e-12 * np.abs(a[z][f]).max(), (kw, z, f)
    lo, hi = {}, {}
    zd.generate_planes(zd.make_params(n, icformat="Zeldovich", stream_factor=8), ps,
                       lambda z, rec: lo.__setitem__(z, rec["d"].copy()) if z in zs else None)
    zd.generate_planes(zd.make_params(2 * n, icformat="Zeldovich", k_cutoff=2.0, stream_factor=16), ps,
                       lambda z, rec: hi.__setitem__(z // 2, rec["d"][::2, ::2].copy()) if (z % 2 == 0 and z // 2 in zs) else None)
    for z in zs:
        assert np.abs(hi[z] - lo[z]).max() < 1e-12 * np.abs(lo[z]).max(), z


@pytest.mark.parametrize("n,R,plt", [(96, 4, True), (160, 4, False), (224, 4, False)])
def test_z_lines_of_eight_times_q_vs_oracle(zd, oracle, ps, opk, n, R, plt):
    """z lines of 8 * Q (round 5: 24, 40, 56 here — one thread per 8-point sub-line in the z transform, `k_zfft_fq<8, 8, Q, NC>`;
    the generator walks 4 z rows per thread) against the oracle, ZA (two residues per pass) and PLT"""
    eig = oracle.synthetic_eigenmodes(24) if plt else None
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97) if plt else {}
    plan = zd.Plan(zd.make_params(n, stream_factor=R, **kw), ps, eig=eig)
    assert plan.store_mode == "fields" and plan.R == R
    plan.close()
    got, _ = _compare(zd, oracle, ps, opk, n, eig=eig, stream_factor=R, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


@pytest.mark.parametrize("n,R,plt", [(288, 4, False), (480, 4, True), (800, 4, False), (864, 4, True), pytest.param(864, 4, False, marks=pytest.mark.slow)])
def test_z_lines_of_eight_times_q_vs_longer_lines(zd, oracle, ps, n, R, plt):
    """z lines of 72, 120, 200, 216 = 8 * {9, 15, 25, 27} (beyond the oracle's plain DFT in the time of a test): stream factor R
    against R = 2 (lines of 16 * 2^k * Q) on sample planes, ZA with a density (six fields) or PLT"""
    zs = (1, n // 2 + 3, n - 2)
    eig = oracle.synthetic_eigenmodes(32)
    for kw, e in (((dict(icformat="RVdoubleZel", qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97), eig),) if plt else
                  ((dict(icformat="RVdoubleZel", qdensity=1), None),)):
        a, b = {}, {}
        sa = zd.generate_planes(zd.make_params(n, stream_factor=2, **kw), ps, lambda z, rec: a.__setitem__(z, rec.copy()) if z in zs else None, eig=e)
        sb = zd.generate_planes(zd.make_params(n, stream_factor=R, **kw), ps, lambda z, rec: b.__setitem__(z, rec.copy()) if z in zs else None, eig=e)
        assert sb["stream_factor"] == R and sorted(a) == sorted(b) == sorted(zs)
        for z in zs:
            for f in ("d", "v"):
                assert np.abs(a[z][f]).max() > 0
                assert np.abs(a[z][f] - b[z][f]).max() <= 1e-12 * np.abs(a[z][f]).max(), (kw, z, f)
        assert abs(sa["density_variance"] - sb["density_variance"]) <= 1e-12 * sa["density_variance"]


@pytest.mark.parametrize("n,Rs,plt", [(864, (6, 18), False), (864, (6,), True), (960, (10, 12, 20), False)])
def test_stream_factors_that_are_not_powers_of_two(zd, oracle, ps, n, Rs, plt):
    """composite PPDs take any even stream factor whose z lines have a composite transform (PPD = 6912 on one GPU: R = 36, 18 passes,
    where the powers of two offer R = 64, 32 passes): sample planes equal to the R = 2 run's, ZA (two residues r, r + R/2 per pass)
    and PLT; the automatic choice picks such a factor when it is the smallest that fits"""
    zs = (1, n // 2 + 7, n - 2)
    kw = dict(icformat="Zeldovich")
    eig = None
    if plt:
        kw = dict(icformat="RVdoubleZel", qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
        eig = oracle.synthetic_eigenmodes(32)
    ref = {}
    zd.generate_planes(zd.make_params(n, stream_factor=2, **kw), ps, lambda z, rec: ref.__setitem__(z, rec.copy()) if z in zs else None, eig=eig)
    for R in Rs:
        got, seen = {}, []

        def take(z, rec):
            seen.append(z)
            if z in zs:
                got[z] = rec.copy()

        st = zd.generate_planes(zd.make_params(n, stream_factor=R, **kw), ps, take, eig=eig)
        assert st["stream_factor"] == R and sorted(seen) == list(range(n))
        for z in zs:
            for f in ("d", "v"):
                if f in ref[z].dtype.names:
                    assert np.abs(got[z][f] - ref[z][f]).max() <= 1e-12 * np.abs(ref[z][f]).max(), (R, z, f)


def test_stream_factor_choice_on_composite_grids(zd):
    """PPD = 6912 on a 288 GB GPU: the smallest even factor whose store fits and whose z lines (192 = 64 * 3) have a transform"""
    import ctypes as C
    L = zd.load_library()
    p = zd.make_params(6912, icformat="RVZel", numblock=64)
    R = L.zd_choose_stream_factor(C.byref(p), 1, (288 - 32) << 30)
    assert R % 2 == 0 and 6912 % R == 0 and R < 64, R


# ---- ZD_Version = 1: legacy mt19937 streams with rejection sampling ----
@pytest.mark.parametrize("seed", [12346, 5489, 0, 2 ** 32 + 7])
def test_v1_stream_words(zd, seed):
    """the workgroup-parallel MT19937 regeneration == the serial generator: numpy's RandomState(int) uses the same
    init_genrand seeding as gsl_rng_set (gsl maps seed 0 to 4357 and keeps the low 32 bits); seed 5489 carries the published
    known answer (10000th word = 4123659995)"""
    w = zd.test_v1_words(seed, 17)
    eff = 4357 if seed == 0 else seed & 0xffffffff
    ref = np.random.RandomState(eff).randint(0, 2 ** 32, size=w.size, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(w, ref)
    if seed == 5489:
        assert int(w[9999]) == 4123659995


@pytest.mark.parametrize("n,kw", [
    (64, dict(numblock=4)),                                              # 16 streams, field store
    (128, dict(numblock=2, stream_factor=4)),                            # 64 streams replayed for every pass
    (64, dict(numblock=8, store_mode="reference", stream_factor=2)),     # 8 streams: slab rows share streams
    (128, dict(numblock=64, store_mode="packed")),                       # 2 streams serving 32 rows each
    (128, dict(numblock=4, k_cutoff=2.0, stream_factor=2)),              # NumBlock -> 8; only the modes inside the cutoff draw
    (64, dict(numblock=4, qdensity=1, fmt="RVZel")),
    (64, dict(numblock=4, corner_modes=1)),
    (64, dict(numblock=4, plt=True)),
    (128, dict(numblock=4, ngpu=2, stream_factor=2)),                    # streams split over the ranks (ky mod G)
    (128, dict(numblock=16, ngpu=4, store_mode="reference")),
])
def test_version1_vs_oracle(zd, oracle, ps, opk, n, kw):
    """ZD_Version = 1 end to end against the oracle's restatement of cgauss<1> / gsl_rng_mt19937 (src/power_spectrum.cpp:18-25,
    310-332; zeldovich.cpp:365-370): the n-th live mode of stream yres takes the n-th accepted pair"""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(32)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, version=1, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))


def test_version1_fixed_power_and_one_mode(zd, oracle, wmap_path):
    ps2 = zd.PowerSpectrum.from_file(wmap_path, 720.0, fix_to_mean=1)
    opk2 = oracle.pk_from_file(wmap_path, 720.0, fix_to_mean=1)
    _compare(zd, oracle, ps2, opk2, 64, version=1, numblock=4)
    _compare(zd, oracle, ps2, opk2, 32, version=1, numblock=2, qonemode=1, one_mode=(1, 2, -3), tie_ok=True)


def test_version1_properties(zd, ps):
    """what the reference documents about version 1 (include/zeldovich.h:27-28, parameters.cpp:129-141): the phases depend on
    ZD_NumBlock, and with NumBlock scaled by k_cutoff an oversampled grid reproduces the coarse one at the shared sites"""
    a = zd.generate(zd.make_params(128, numblock=4, version=1), ps)["records"]["d"]
    b = zd.generate(zd.make_params(256, numblock=4, version=1, k_cutoff=2.0, stream_factor=2), ps)["records"]["d"][::2, ::2, ::2]
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    c = zd.generate(zd.make_params(128, numblock=2, version=1), ps)["records"]["d"]
    assert np.abs(a - c).max() > 0.1 * np.abs(a).max()
    d = zd.generate(zd.make_params(128, numblock=4), ps)["records"]["d"]      # version 2: other phases
    assert np.abs(a - d).max() > 0.1 * np.abs(a).max()


@pytest.mark.parametrize("ngpu", [1, 2])
def test_version1_with_fnl(zd, oracle, ps, wmap_path, ngpu):
    """ZD_f_NL on the legacy streams: the phi round draws with cgauss<1>, the second pass takes D from PhiK"""
    import ctypes as C
    n, fnl, ns, om = 64, 2.0e4, 0.96, 0.31
    opk = oracle.pk_from_file(wmap_path, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(opk), ns)
    got = zd.generate(zd.make_params(n, numblock=4, version=1, f_NL=fnl, n_s=ns, Omega_M=om, ngpu=ngpu), ps)
    ref = oracle.run(oracle.make_params(n, numblock=4, version=1, f_NL=fnl, n_s=ns, Omega_M=om), opk)
    for f in ("d", "v"):
        assert _rel(got["records"][f], ref["records"][f]) < TOL


# ---- any even PPD: convolution (Bluestein) transforms on the power-of-two engine (zd_kernels_any.hip) ----
@pytest.mark.parametrize("n", [10, 50, 100, 125, 250, 1000, 1001, 2500, 5000, 8190])
@pytest.mark.parametrize("kind", [0, 1])
def test_fft_lines_any_length(zd, n, kind):
    """arbitrary lengths (even, odd, prime factors 5, 7, 11, 13 ...) against numpy; ragged tiles (7 lines)"""
    rng = np.random.default_rng(n + kind)
    lines = 7
    x = rng.standard_normal((lines, n)) + 1j * rng.standard_normal((lines, n))
    got = zd.test_fft(x, kind)
    ref = np.fft.ifft(x, axis=1) * n
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-14, err


@pytest.mark.parametrize("n,kw", [
    (50, dict()),                                               # 2 * 5^2: z lines of 50 (generator walks 2 rows)
    (100, dict(stream_factor=2, fmt="RVZel")),
    (70, dict(qdensity=1)),                                     # 2 * 5 * 7, density plane too
    (110, dict(qdensity=2)),                                    # 2 * 5 * 11, density only
    (60, dict(plt=True, fmt="RVdoubleZel")),                    # PLT + rescale: four arrays
    (130, dict(k_cutoff=2.0, stream_factor=1)),                 # 2 * 5 * 13
    (96, dict(qdensity=1, stream_factor=4)),                    # a 2^a 3^b size whose option (density) the composite kernels lack
    (100, dict(corner_modes=1, k_cutoff=2.0)),                  # Nyquist-plane modes alive: the reference arrays are the only exact store
    (100, dict(version=1, numblock=4)),                         # legacy streams
    (100, dict(stream_factor=4, fmt="Zeldovich")),              # z lines of 25: an odd length (generator walks one row)
    (126, dict(stream_factor=2)),                               # 2 * 3^2 * 7: z lines of 63
    (50, dict(stream_factor=5)),                                # a stream factor that is not a power of two: z lines of 10
    (90, dict(stream_factor=3, fmt="RVZel")),                   # 2 * 3^2 * 5, three residue passes
    (100, dict(stream_factor=25)),                              # z lines of 4
    # several GPUs (round 5): the convolution kernels run on one rank, so the GPUs share such a job as pass groups — the library picks a
    # divisor of PPD that deals its passes out over them (zd_choose_pass_groups, convolution_job)
    (100, dict(ngpu=2, pass_groups=0, fmt="RVZel")),           # R = 2: one pass per GPU
    (90, dict(ngpu=2, pass_groups=0, qdensity=1)),             # 2 * 3^2 * 5
    (96, dict(ngpu=4, pass_groups=0, corner_modes=1, k_cutoff=2.0)),  # a composite size whose option needs the reference arrays
])
def test_any_even_ppd_vs_oracle(zd, oracle, ps, opk, n, kw):
    """PPD with prime factors other than 2 and 3 (the reference plans any length with FFTW, src/zeldovich.cpp:61-66; its only
    conditions are an even ppd divisible by NumBlock): reference arrays transformed as convolutions on the power-of-two
    engine, every record against the oracle (whose non-power-of-two path is a plain DFT)"""
    kw = dict(kw)
    fmt = kw.pop("fmt", "RVdoubleZel")
    eig = None
    if kw.pop("plt", False):
        eig = oracle.synthetic_eigenmodes(24)
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, f_cluster=0.97)
    got, _ = _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, **kw)
    assert sorted(got["planes_seen"]) == list(range(n))
