"""CPU tests that PIN the oracle (oracle/zd_oracle.c):
  * against known-answer vectors generated from the reference's own pcg64 / SplineFunction object
    code (tests/golden/*.json, made by tests/golden/make_golden.py) and the vectors recorded from a
    reference run in SURVEY.md §8(c);
  * live against oracle/_ref when that library exists (build container only);
  * against an independent numpy formulation of the mode cube + FFT;
  * through the invariants the reference documents (README: NumBlock independence, oversampling,
    fix-to-mean phases) and its RNG-distance self check (src/zeldovich.cpp:478).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, WMAP

M64 = 2 ** 64 - 1


def _pcg(oracle, seed):
    g = oracle.Pcg()
    oracle.lib().zdo_pcg_seed(C.byref(g), seed & M64)
    return g


def test_pcg_known_answers(oracle):
    L = oracle.lib()
    kat = json.load(open(os.path.join(GOLDEN, "pcg_kat.json")))
    for e in kat["seeds"]:
        g = _pcg(oracle, e["seed"])
        assert [hex(g.hi), hex(g.lo)] == e["state0"]
        assert [hex(L.zdo_pcg_next(C.byref(g))) for _ in range(6)] == e["draws"]
        assert [hex(g.hi), hex(g.lo)] == e["after_draws"]
        g = _pcg(oracle, e["seed"])
        L.zdo_pcg_advance(C.byref(g), 0, 2 * 65536 * 65536)
        assert [hex(g.hi), hex(g.lo)] == e["after_plane_advance"]
        g = _pcg(oracle, e["seed"])
        d = (1 << 70) + 12345
        L.zdo_pcg_advance(C.byref(g), d >> 64, d & M64)
        assert [hex(g.hi), hex(g.lo)] == e["after_2p70_12345"]
    # counter addressing of modes
    p = oracle.make_params(4096)
    pk = oracle.pk_from_powerlaw(-1.0, 720.0)
    r = (C.c_uint64 * 2)()
    D = (C.c_double * 2)()
    for m in kat["modes"]:
        kx, ky, kz = m["k"]
        L.zdo_mode_draw(C.byref(p), C.byref(pk), kx, ky, kz, r, D)
        assert [hex(r[0]), hex(r[1])] == m["r"], m


def test_survey_golden_vectors(oracle):
    """values recorded from a run of the reference binary, seed 12346"""
    L = oracle.lib()
    g = _pcg(oracle, 12346)
    assert (g.hi << 64 | g.lo) == 0x78d867e7fdc277d9125704dedb20013c
    assert [L.zdo_pcg_next(C.byref(g)) for _ in range(4)] == [0xb9a1e6ce3f08d3b0, 0xb81440d0f52635bc,
                                                              0xc4ea9ea5313d5238, 0x5362ce3b88196450]
    gold = {(3, 5, 7): (42950590470, 0x73da5750b0db04d6, 0xcd339d3f2f327ebd, 0.45255037040702112, 0.80156882088267345),
            (-3, 5, 7): (42950721530, 0x0a1dac63ed2b85b8, 0xbb7c44248c0563e1, None, None),
            (3, 5, -7): (51538690054, 0x777373dcf4c2866b, 0xc3988c4a2ab1f703, None, None),
            (-9, 1, -2): (17179738094, 0xd513f641bab2a889, 0x88df5069d40b5329, None, None),
            (1, 0, 2): (262146, 0x95e443207a26a8b2, 0x8e5348004a6e721b, None, None),
            (0, 2, 0): (17179869184, 0xc658b8961a1051d7, 0x5f97cac5840856c6, None, None)}
    p = oracle.make_params(128)
    pk = oracle.pk_from_powerlaw(-1.0, 720.0)
    r = (C.c_uint64 * 2)()
    D = (C.c_double * 2)()
    for (kx, ky, kz), (c, r1, r2, u1, u2) in gold.items():
        assert 2 * ((ky * 65536 + (kz & 65535)) * 65536 + (kx & 65535)) == c
        L.zdo_mode_draw(C.byref(p), C.byref(pk), kx, ky, kz, r, D)
        assert (r[0], r[1]) == (r1, r2)
        if u1 is not None:
            assert L.zdo_u01(r1) == u1 and L.zdo_u01(r2) == u2


def test_u01_edges(oracle):
    L = oracle.lib()
    assert L.zdo_u01(M64) == 1.0
    assert L.zdo_u01(0) == 2.0 ** -64
    assert L.zdo_u01(M64 - 1) == 1.0  # 2^64-1 rounds to 2^64 in double
    assert 0.0 < L.zdo_u01(12345) <= 1.0


def test_spline_known_answers(oracle):
    L = oracle.lib()
    kat = json.load(open(os.path.join(GOLDEN, "spline_kat.json")))
    tab = np.loadtxt(WMAP)
    probes = np.array([float.fromhex(v) for v in kat["probes"]])
    for key, perm in (("val", None), ("val_permuted_nodes", kat["perm"])):
        x, y = np.log(tab[:, 0]), np.log(tab[:, 1])
        if perm is not None:
            x, y = x[perm], y[perm]
        x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
        y2 = np.zeros_like(x)
        L.zdo_spline_build(len(x), x.ctypes.data, y.ctypes.data, y2.ctypes.data)
        if perm is None:
            assert np.all(np.diff(x) > 0)
        # NB: for shuffled nodes the reference's shell sort compares x[j-inc] of the ZERO-based array
        # (spline_function.h:95) and does not fully sort; the oracle restates that literally, so the
        # known answers below (taken from the reference object code) still have to match.
        want = np.array([float.fromhex(v) for v in kat[key]])
        got = np.array([L.zdo_spline_val(len(x), x.ctypes.data, y.ctypes.data, y2.ctypes.data, float(v)) for v in probes])
        # same arithmetic, possibly different FMA contraction between the two builds: <= 4 ulp
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)) < 1e-15


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libzd_ref.so")),
                    reason="oracle/_ref is only built where /root/reference is mounted")
def test_pcg_against_reference_object_code(oracle):
    R, L = oracle.ref(), oracle.lib()
    rng = np.random.default_rng(3)
    u64 = C.c_uint64
    for _ in range(50):
        seed = int(rng.integers(0, 2 ** 63))
        hi, lo = u64(), u64()
        R.ref_pcg_seed(seed, C.byref(hi), C.byref(lo))
        g = _pcg(oracle, seed)
        assert (g.hi, g.lo) == (hi.value, lo.value)
        delta = int(rng.integers(0, 2 ** 62)) << int(rng.integers(0, 40))
        R.ref_pcg_advance(C.byref(hi), C.byref(lo), delta >> 64, delta & M64)
        L.zdo_pcg_advance(C.byref(g), delta >> 64, delta & M64)
        assert (g.hi, g.lo) == (hi.value, lo.value)
        out = (u64 * 5)()
        a = (hi.value, lo.value)
        R.ref_pcg_draw(C.byref(hi), C.byref(lo), 5, out)
        assert [L.zdo_pcg_next(C.byref(g)) for _ in range(5)] == list(out)
        # distance (the reference's operator-, used by its RNG self check)
        ga = oracle.Pcg(a[0], a[1])
        assert L.zdo_pcg_distance(C.byref(ga), C.byref(g)) == R.ref_pcg_distance(a[0], a[1], hi.value, lo.value) == 5


# ---- independent formulation (numpy) -----------------------------------------
def _pcg_py(seed):
    MULT = 0x2360ed051fc65da44385df649fccf645
    INC = 0x5851f42d4c957f2d14057b7ef767814f
    M = (1 << 128) - 1
    s = (((seed & M64) + INC) * MULT + INC) & M

    def adv(s, d):
        am, ap, cm, cp = 1, 0, MULT, INC
        while d:
            if d & 1:
                am = am * cm & M
                ap = (ap * cm + cp) & M
            cp = (cm + 1) * cp & M
            cm = cm * cm & M
            d >>= 1
        return (am * s + ap) & M

    def out(s):
        x = ((s >> 64) ^ s) & M64
        r = s >> 122
        return ((x >> r) | (x << ((64 - r) & 63))) & M64

    return s, adv, out, lambda s: (s * MULT + INC) & M


class _IndependentPk:
    """P(k) of PowerSpectrum::power (src/power_spectrum.cpp:225-261) WITHOUT any oracle code: scipy's natural cubic
    spline through (ln k, ln P) of the table file (the last / first cubic piece extrapolates, as SplineFunction::val
    does with its clamped segment index) and an adaptive quadrature for sigma_R (the reference uses Romberg to 1e-6 on
    [0, 10], src/power_spectrum.cpp:60-128 — so the two normalisations agree to ~1e-6, not to rounding)."""

    def __init__(self, path, boxsize, Pk_norm=8.0, Pk_sigma=0.0210839935761):
        from scipy.integrate import quad
        from scipy.interpolate import CubicSpline
        tab = np.loadtxt(path)
        self.spl = CubicSpline(np.log(tab[:, 0]), np.log(tab[:, 1]), bc_type="natural", extrapolate=True)

        def integrand(k):
            x = k * Pk_norm
            w = 1 - x * x / 10.0 if x <= 1e-3 else 3.0 * (np.sin(x) - x * np.cos(x)) / x ** 3
            return 0.5 / np.pi ** 2 * k * k * w * w * self.shape(k)

        var = quad(integrand, 0.0, 10.0, limit=2000, epsabs=0, epsrel=1e-11,
                   points=list(np.linspace(0.01, 9.99, 400)))[0]
        self.norm = (Pk_sigma / np.sqrt(var)) ** 2 / boxsize ** 3

    def shape(self, k):
        return float(np.exp(self.spl(np.log(k)))) if k > 0 else 0.0

    def __call__(self, k):
        return self.shape(k) * self.norm


def _independent_eigenmode(eig, n, kx, ky, kz):
    """get_eigenmode / interp_eigmode (src/zeldovich.cpp:154-276) in numpy, written from the reference's rules, not from
    the oracle: per-axis (lower corner, upper corner, fraction) with the "do not interpolate across the Nyquist seam"
    bump (:176-183) and the wrap of the upper corner (:194-198); tensor-product weights; e_z sign; renormalise;
    e k^2/(k.e)."""
    ep = eig.shape[0]

    def axis(ik):
        if ep % n == 0:
            return [(ik * (ep // n), 1.0)]
        f = ep / n * ik
        if ep // 2 < f < ep // 2 + 1:
            f = np.floor(f + 1)
        lo = int(f)
        hi = (lo + 1) % ep
        t = f - lo
        return [(lo, 1.0 - t)] + ([(hi, t)] if t != 0 else [])

    ikx, iky, ikz = kx % n, ky % n, kz % n
    if ikz > n // 2:
        ikz = n - ikz
    e = np.zeros(4)
    for ix, wx in axis(ikx):
        for iy, wy in axis(iky):
            for iz, wz in axis(ikz):
                e += wx * wy * wz * eig[ix, iy, iz]
    e[2] *= -1.0 if kz < 0 else 1.0
    e[:3] /= np.sqrt(e[0] ** 2 + e[1] ** 2 + e[2] ** 2)
    k2 = float(kx * kx + ky * ky + kz * kz)
    with np.errstate(divide="ignore", invalid="ignore"):
        norm = k2 / (kx * e[0] + ky * e[1] + kz * e[2])
    if k2 == 0 or not np.isfinite(norm):
        norm = 0.0
    return norm * e[0], norm * e[1], norm * e[2], e[3]


def _independent_cube(n, power, seed=12346, k_cutoff=1.0, boxsize=720.0, eig=None, f_cluster=1.0, rescale=None):
    """packed arrays [a][ky][kz][kx] built straight from Appendix B1/B2/B3 (no blocks, no twins-by-copy);
    `power(k)` and the eigenmode lookup are numpy/scipy code, nothing here calls the oracle"""
    fund = 2 * np.pi / boxsize
    nyq = np.pi / (boxsize / n)
    kmax = int((n // 2) / k_cutoff + .5)
    k2cut = nyq * nyq / (k_cutoff * k_cutoff)
    s0, adv, out, step = _pcg_py(seed)
    na = 4 if eig is not None else 2
    cube = np.zeros((na, n, n, n), dtype=np.complex128)
    fk = lambda i: i - n if i > n // 2 else i

    def draw(kx, ky, kz):
        if max(abs(kx), abs(ky), abs(kz)) == kmax:
            return 0j
        k2 = (kx * kx + ky * ky + kz * kz) * fund * fund
        if k2 >= k2cut:
            return 0j
        c = 2 * ((ky * 65536 + (kz & 65535)) * 65536 + (kx & 65535))
        s = step(adv(s0, c))
        r1 = out(s)
        r2 = out(step(s))
        u = lambda r: 1.0 if r == M64 else float(r + 1) * 2.0 ** -64
        P = power(float(np.sqrt(k2)))
        amp = np.sqrt(-P * np.log(u(r1)))
        th = 2 * np.pi * u(r2)
        return amp * (np.cos(th) + 1j * np.sin(th))

    def fields(kx, ky, kz, D):
        k2 = (kx * kx + ky * ky + kz * kz) * fund * fund or 1.0
        if eig is None:
            e, lam, f, resc = (kx, ky, kz), 1.0, 1.0, 1.0
        else:
            ev = _independent_eigenmode(eig, n, kx, ky, kz)
            e, lam = (ev[0], ev[1], ev[2]), ev[3]
            f = (np.sqrt(1 + 24 * lam * f_cluster) - 1) / 4
            resc = 1.0 if rescale is None else rescale ** ((np.sqrt(1 + 24 * f_cluster) - 1) / 4 - f)
        F = [1j * resc * e[j] * fund / k2 * D for j in range(3)]
        return F, f

    def put(ix, iy, iz, D, F, f, conj):
        cj = (lambda v: np.conj(v)) if conj else (lambda v: v)
        cube[0, iy, iz, ix] = cj(D) + 1j * cj(F[0])
        cube[1, iy, iz, ix] = cj(F[1]) + 1j * cj(F[2])
        if na == 4:
            cube[2, iy, iz, ix] = 1j * cj(F[0] * f)
            cube[3, iy, iz, ix] = cj(F[1] * f) + 1j * cj(F[2] * f)

    for iy in range(0, n // 2):
        for iz in range(n):
            for ix in range(n):
                kx, ky, kz = fk(ix), iy, fk(iz)
                if iy == 0:
                    winner = (0 < iz < n // 2) or (iz == 0 and 0 < ix < n // 2)
                    if not winner:
                        continue
                D = draw(kx, ky, kz)
                if D == 0:
                    continue
                F, f = fields(kx, ky, kz, D)
                put(ix, iy, iz, D, F, f, False)
                put((n - ix) % n, (n - iy) % n, (n - iz) % n, D, F, f, True)
    return cube


def test_blockarray_known_answers(oracle):
    """StoreBlock / LoadBlock of the oracle against images produced by the REFERENCE's BlockArray object code
    (tests/golden/blockarray_kat.json; src/block_array.cpp:387-414,466-504, include/block_array.h:33-34)"""
    import hashlib
    kat = json.load(open(os.path.join(GOLDEN, "blockarray_kat.json")))
    assert len(kat["cases"]) >= 6
    for c in kat["cases"]:
        arr, slabs = oracle.blockarray_roundtrip(oracle.lib().zdo_blockarray_roundtrip, c["ppd"], c["numblock"],
                                                 c["narray"], fill=c["fill"])
        assert hashlib.sha256(arr.tobytes()).hexdigest() == c["arr_sha256"], c
        assert hashlib.sha256(slabs.tobytes()).hexdigest() == c["slabs_sha256"], c
        if "arr_re" in c:
            assert np.array_equal(arr[:, 0], np.array(c["arr_re"], dtype=np.float64))
            assert np.array_equal(slabs[:, 0], np.array(c["slabs_re"], dtype=np.float64))
            assert np.array_equal(arr[:, 1], -arr[:, 0] - 0.25)
        # what the layout means (block_array.h:33-34 + the y shift of LoadBlock): element (y, a, z, x) of the z stage
        # lands in plane z, row yshift(y) of the xy stage
        ppd, na = c["ppd"], c["narray"]
        inp = oracle.blockarray_input(ppd, c["numblock"], na)[:, 0].reshape(ppd, na, ppd, ppd)  # [y][a][z][x]
        got = slabs[:, 0].reshape(ppd, na, ppd, ppd)  # [z][a][y][x]
        for y in range(ppd):
            ys = y if y < ppd // 2 else (ppd // 2 if y + 1 == ppd else y + 1)
            assert np.array_equal(got[:, :, ys, :], inp[y].transpose(1, 0, 2))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libzd_ref.so")),
                    reason="oracle/_ref only exists in the build container")
def test_blockarray_against_reference_object_code(oracle):
    for ppd, nb, na in [(8, 2, 2), (16, 4, 4), (24, 2, 2), (12, 6, 1)]:
        a, b = oracle.blockarray_roundtrip(oracle.ref().ref_blockarray_roundtrip, ppd, nb, na)
        c, d = oracle.blockarray_roundtrip(oracle.lib().zdo_blockarray_roundtrip, ppd, nb, na)
        assert np.array_equal(a, c) and np.array_equal(b, d)


@pytest.mark.parametrize("plt,box", [(False, 720.0), (True, 720.0), (False, 3.0), (True, 6.0)])
def test_oracle_vs_independent_formulation(oracle, plt, box):
    """box = 3 / 6 Mpc/h: k_Nyquist = 16.8 / 8.4 h/Mpc, i.e. most modes lie beyond the last node of wmap1new.pow
    (k = 5.13) where SplineFunction::val extrapolates with its last cubic — the regime PPD = 2048 ... 8192 run in"""
    n = 16
    pk = oracle.pk_from_file(WMAP, box)
    eig = oracle.synthetic_eigenmodes(12) if plt else None
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0, f_cluster=0.95) if plt else {}
    p = oracle.make_params(n, numblock=4, boxsize=box, **kw)
    cube = oracle.mode_cube(p, pk, eig=eig, eig_ppd=0 if eig is None else eig.shape[0])
    ipk = _IndependentPk(WMAP, box)
    ind = _independent_cube(n, ipk, boxsize=box, eig=eig, f_cluster=0.95 if plt else 1.0,
                            rescale=(1 / 6.0) / (1 / 50.0) if plt else None)
    # sigma_R: Romberg to 1e-6 (reference) vs adaptive quadrature (here): one global amplitude factor ~1e-6 from 1
    amp = np.sqrt(pk.normalization / ipk.norm)
    assert abs(amp - 1.0) < 3e-6, amp
    ind *= amp
    scale = np.abs(ind).max()
    assert np.abs(cube - ind).max() / scale < 1e-13
    out = oracle.run(p, pk, eig=eig, eig_ppd=0 if eig is None else eig.shape[0], want_planes=True)
    ref = (np.fft.ifftn(ind, axes=(1, 2, 3)) * n ** 3).transpose(2, 0, 1, 3)  # [z][a][y][x]
    assert np.abs(out["planes"] - ref).max() / np.abs(ref).max() < 1e-12  # n^3 = 4096 terms per sample
    # records are the unpacked planes (src/output.cpp:93-141)
    r = out["records"]
    assert np.array_equal(r["ijk"][3, 5, 7], [3, 5, 7])
    assert np.allclose(r["d"][..., 2], ref[:, 0].imag, rtol=0, atol=1e-12 * np.abs(ref).max())
    assert np.allclose(r["d"][..., 1], ref[:, 1].real, rtol=0, atol=1e-12 * np.abs(ref).max())
    assert np.allclose(r["d"][..., 0], ref[:, 1].imag, rtol=0, atol=1e-12 * np.abs(ref).max())
    if plt:
        assert np.allclose(r["v"][..., 2], ref[:, 2].imag, rtol=0, atol=1e-12 * np.abs(ref).max())
        assert np.abs(ref[:, 2].real).max() < 1e-12 * np.abs(ref).max()  # slab[2] real part is identically 0
    assert abs(out["density_variance"] - np.sum(ref[:, 0].real ** 2)) < 1e-12 * out["density_variance"]


def test_numblock_invariance(oracle):
    """README: v2 output does not depend on ZD_NumBlock (nor on the thread count)"""
    pk = oracle.pk_from_file(WMAP, 720.0)
    a = oracle.run(oracle.make_params(32, numblock=2, nthreads=1), pk)
    b = oracle.run(oracle.make_params(32, numblock=8, nthreads=3), pk)
    assert np.abs(a["records"]["d"] - b["records"]["d"]).max() < 1e-14
    assert np.array_equal(a["records"]["ijk"], b["records"]["ijk"])


def test_oversampling_invariance(oracle):
    """README: PPD=2N with k_cutoff=2 sub-sampled x2 equals PPD=N with k_cutoff=1"""
    pk = oracle.pk_from_file(WMAP, 720.0)
    lo = oracle.run(oracle.make_params(16, numblock=2), pk)
    hi = oracle.run(oracle.make_params(32, numblock=2, k_cutoff=2.0), pk)
    assert np.abs(hi["records"]["d"][::2, ::2, ::2] - lo["records"]["d"]).max() < 1e-14


def test_fix_to_mean_keeps_phases(oracle):
    pk = oracle.pk_from_file(WMAP, 720.0)
    pkf = oracle.pk_from_file(WMAP, 720.0, fix_to_mean=1)
    a = oracle.mode_cube(oracle.make_params(16), pk)[0]
    b = oracle.mode_cube(oracle.make_params(16), pkf)[0]
    # density part of array 0 at a +k mode: same phase, amplitude sqrt(P)
    L = oracle.lib()
    fund = 2 * np.pi / 720.0
    for (kx, ky, kz) in [(1, 2, 3), (-2, 1, 0), (0, 3, -1)]:
        D = (C.c_double * 2)()
        r = (C.c_uint64 * 2)()
        L.zdo_mode_draw(C.byref(oracle.make_params(16)), C.byref(pk), kx, ky, kz, r, D)
        Df = (C.c_double * 2)()
        L.zdo_mode_draw(C.byref(oracle.make_params(16)), C.byref(pkf), kx, ky, kz, r, Df)
        d, df = complex(D[0], D[1]), complex(Df[0], Df[1])
        assert abs(np.angle(d) - np.angle(df)) < 1e-14
        k = np.sqrt(kx * kx + ky * ky + kz * kz) * fund
        assert abs(abs(df) - np.sqrt(L.zdo_power(C.byref(pkf), k))) < 1e-15 * abs(df) + 1e-300
    assert a.shape == b.shape


def test_one_mode_is_a_plane_wave(oracle):
    pk = oracle.pk_from_file(WMAP, 720.0)
    n = 16
    out = oracle.run(oracle.make_params(n, qonemode=1, one_mode=(2, 3, -1)), pk, want_planes=True)
    dens = out["planes"][:, 0].real  # [z][y][x]
    spec = np.fft.fftn(dens)
    mag = np.abs(spec)
    top = np.argwhere(mag > 1e-9 * mag.max())
    assert len(top) == 2  # +k and -k only


def test_fnl_path_vs_independent_numpy(oracle):
    """local primordial non-Gaussianity (src/zeldovich.cpp:377-400,699-790,945-960): phi = D/M ->
    phi + f_NL phi^2 in real space -> back to D = phi M, checked against a direct numpy evaluation"""
    n, fnl, ns, om, zi = 16, 3.0e4, 0.96, 0.3, 49.0
    pk = oracle.pk_from_file(WMAP, 720.0)
    oracle.lib().zdo_pk_set_primordial(C.byref(pk), ns)
    L = oracle.lib()
    base = oracle.mode_cube(oracle.make_params(n, qdensity=2), pk)[0]      # D(k) [ky][kz][kx]
    fund = 2 * np.pi / 720.0
    idx = np.arange(n)
    ksig = np.where(idx > n // 2, idx - n, idx)
    KY, KZ, KX = np.meshgrid(ksig, ksig, ksig, indexing="ij")
    k2 = (KX * KX + KY * KY + KZ * KZ) * fund * fund
    kmag = np.sqrt(k2)
    Tk = np.array([L.zdo_infer_Tk(C.byref(pk), float(k)) for k in kmag.ravel()]).reshape(kmag.shape)
    k2s = np.where(k2 == 0, 1.0, k2)
    growth, c, H0 = 1.0 / (1 + zi), 299792.458, 100.0
    M = 2.0 * growth * c * c * Tk * k2s / (3.0 * om * H0 * H0)
    phi = np.fft.ifftn(base / M).real * n ** 3            # imag part is rounding (Hermitian cube)
    g = (phi + fnl * phi * phi) / n ** 3
    Dng = np.fft.fftn(g) * M
    Dng[0, 0, 0] = 0.0
    Dng[n // 2, :, :] = 0.0                                # the Nyquist row is never generated
    dens = np.fft.ifftn(Dng).real * n ** 3                 # [y][z][x]
    out = oracle.run(oracle.make_params(n, numblock=2, qdensity=2, f_NL=fnl, n_s=ns, Omega_M=om, z_initial=zi), pk,
                     want_planes=True)
    got = out["planes"][:, 0].real                         # [z][y][x]
    assert np.abs(got - dens.transpose(1, 0, 2)).max() < 1e-11 * np.abs(dens).max()
    # and the non-Gaussian term really changes the field
    lin = oracle.run(oracle.make_params(n, numblock=2, qdensity=2), pk, want_planes=True)["planes"][:, 0].real
    assert np.abs(got - lin).max() > 1e-4 * np.abs(lin).max()


# ---- ZD_Version = 1 (legacy streams): gsl_rng_mt19937 + rejection Box-Muller ------------------------------------------
class _Mt(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int)]


def test_mt19937_known_answers(oracle):
    """GSL is not under /root/reference; gsl_rng_mt19937 is MT19937 with the 2002 initialisation (gsl rng/mt.c).  Published
    known answer: seed 5489 -> 10000th word 4123659995 (the value ISO C++ requires of std::mt19937); numpy's legacy
    RandomState(int) seeds with the same init_genrand; gsl_rng_set maps seed 0 to 4357."""
    L = oracle.lib()
    L.zdo_mt_seed.argtypes = [C.POINTER(_Mt), C.c_ulong]
    L.zdo_mt_next.restype = C.c_uint32
    L.zdo_mt_next.argtypes = [C.POINTER(_Mt)]
    L.zdo_mt_uniform.restype = C.c_double
    L.zdo_mt_uniform.argtypes = [C.POINTER(_Mt)]
    g = _Mt()
    L.zdo_mt_seed(C.byref(g), 5489)
    w = [L.zdo_mt_next(C.byref(g)) for _ in range(10000)]
    assert w[0] == 3499211612 and w[9999] == 4123659995
    for seed, eff in ((12346, 12346), (0, 4357), (2 ** 32 + 7, 7)):
        L.zdo_mt_seed(C.byref(g), seed)
        ref = np.random.RandomState(eff).randint(0, 2 ** 32, size=2000, dtype=np.uint64)
        assert [L.zdo_mt_next(C.byref(g)) for _ in range(2000)] == [int(v) for v in ref]
    L.zdo_mt_seed(C.byref(g), 12346)
    assert L.zdo_mt_uniform(C.byref(g)) == 3990012703 / 4294967296.0  # gsl_rng_uniform: word / 2^32 in [0, 1)


def test_version1_vs_independent_numpy(oracle):
    """the `ver == 1` branch (src/zeldovich.cpp:365-370, power_spectrum.cpp:18-25,310-332) written independently: stream yres
    = RandomState(seed + yres) serves rows yres + yblock*block in yblock order; inside a row z-major / x-minor, only modes
    that survive the zero rule draw; pairs of uniforms 2u - 1 until 0 < r2 < 1; D = (p1, p2) sqrt(-P ln r2 / r2)"""
    n, nb, box, seed = 16, 4, 720.0, 12346
    pk = oracle.pk_from_file(WMAP, box)
    p = oracle.make_params(n, numblock=nb, boxsize=box, seed=seed, version=1)
    cube = oracle.mode_cube(p, pk)  # [a][ky][kz][kx], a = 0: D + i F_x ... use the density row below
    block, half = n // nb, n // 2
    fund = 2 * np.pi / box
    k2cut = (np.pi / (box / n)) ** 2
    streams = [np.random.RandomState(seed + i) for i in range(block)]
    L = oracle.lib()
    D = np.zeros((n, n, n), dtype=np.complex128)  # [ky][kz][kx], half-space rows
    for yblock in range(nb // 2):
        for yres in range(block):
            y = yres + yblock * block
            rs = streams[yres]
            for z in range(n):
                kz = z - n if z > half else z
                for x in range(n):
                    kx = x - n if x > half else x
                    k2 = (kx * kx + y * y + kz * kz) * fund * fund
                    if max(abs(kx), abs(y), abs(kz)) == half or k2 >= k2cut:
                        continue
                    while True:
                        u = rs.randint(0, 2 ** 32, size=2, dtype=np.uint64) / 4294967296.0
                        p1, p2 = u[0] * 2.0 - 1.0, u[1] * 2.0 - 1.0
                        r2 = p1 * p1 + p2 * p2
                        if 0.0 < r2 < 1.0:
                            break
                    q = np.sqrt(-L.zdo_power(C.byref(pk), float(np.sqrt(k2))) * np.log(r2) / r2)
                    D[y, z, x] = complex(p1 * q, p2 * q)
    # density-only run: the packed cube holds D itself for the half-space rows
    pd = oracle.make_params(n, numblock=nb, boxsize=box, seed=seed, version=1, qdensity=2)
    cd = oracle.mode_cube(pd, pk)[0]
    for y in range(1, half):
        assert np.abs(cd[y] - D[y]).max() <= 1e-15 * np.abs(D).max(), y
    assert cube.shape[0] == 2


def test_version1_properties(oracle):
    """include/zeldovich.h:27-28: version-1 phases depend on ZD_NumBlock; parameters.cpp:129-141: NumBlock scaled by k_cutoff
    keeps an oversampled grid on the coarse grid's phases"""
    pk = oracle.pk_from_file(WMAP, 720.0)
    a = oracle.run(oracle.make_params(32, numblock=4, version=1), pk)["records"]["d"]
    b = oracle.run(oracle.make_params(64, numblock=4, version=1, k_cutoff=2.0), pk)["records"]["d"][::2, ::2, ::2]
    assert np.abs(a - b).max() < 1e-14
    c = oracle.run(oracle.make_params(32, numblock=2, version=1), pk)["records"]["d"]
    assert np.abs(a - c).max() > 0.1 * np.abs(a).max()
    d = oracle.run(oracle.make_params(32, numblock=4, version=1, nthreads=1), pk)["records"]["d"]
    assert np.array_equal(a, d)  # one stream per yres: no dependence on the thread count


# ---- direct-summation fixture (tests/golden/direct_sum.json) --------------------------------------------------------------
def test_direct_sum_fixture_small_cases_equal_the_full_oracle_run(oracle):
    """tests/golden/direct_sum.json is what the GPU suite checks the FULL-SIZE runs against (test_gpu_direct_sum.py).  Its
    small cases are re-derived here two ways: zdo_direct_sum again (the fixture is what the committed generator produces)
    and the oracle's complete path — LoadPlane, BlockArray, FFTs, WriteParticlesSlab — whose records at the same sites must
    equal the mode-by-mode sums."""
    import json
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_direct_sum", os.path.join(GOLDEN, "make_direct_sum.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fx = json.load(open(os.path.join(GOLDEN, "direct_sum.json")))
    assert set(fx) == set(gen.CASES)
    opk = oracle.pk_from_file(WMAP, 720.0)
    for name, c in fx.items():
        assert [tuple(s) for s in c["sites"]] == gen.sites_for(c["ppd"]), name
        assert c["params"] == gen.CASES[name]["kw"] and c["eig_ppd"] == gen.CASES[name].get("eig", 0), name
        if c["ppd"] > 256:
            continue
        n = c["ppd"]
        eig = oracle.synthetic_eigenmodes(c["eig_ppd"]) if c["eig_ppd"] else None
        p = oracle.make_params(n, numblock=2, **c["params"])
        again = oracle.direct_sum(p, opk, c["sites"], eig=eig)
        want = np.array(c["values"])
        assert np.abs(again - want).max() <= 1e-13 * np.abs(want).max(), name
        ref = oracle.run(p, opk, eig=eig, eig_ppd=c["eig_ppd"])["records"]
        scale = np.abs(ref["d"]).max()
        for (z, y, x), w in zip(c["sites"], want):
            assert np.abs(ref["d"][z, y, x][::-1] - w[:3]).max() <= 1e-12 * scale, (name, z, y, x)
            assert np.abs(ref["v"][z, y, x][::-1] - w[3:6]).max() <= 1e-12 * scale, (name, z, y, x)
