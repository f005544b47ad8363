"""Soak run of the parity comparison: many seeded random option combinations, each against the oracle (ZD_RUN_SLOW=1; not part of the
default GPU suite).  Wider than test_randomised_option_sweep_vs_oracle: composite grids (2^a 3 / 5 / 7), densities with PLT, several
ranks and pass groups, one-mode runs.  ZD_SOAK_SEED / ZD_SOAK_TRIALS choose the sequence; every failing trial is listed, not only the first."""
import os

import numpy as np
import pytest

from test_gpu_parity import _compare

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def zd():
    import zeldovich_plt_amd.api as api
    api.load_library()
    return api


def _draw(rng, oracle):
    n = int(rng.choice([64, 64, 96, 128, 128, 160, 192, 224, 256]))
    comp = n & (n - 1) != 0
    plt = bool(rng.integers(0, 2)) and n <= 192
    kw = dict(seed=int(rng.integers(1, 2 ** 31 - 1)), k_cutoff=float(rng.choice([1.0, 1.0, 1.0, 1.5, 2.0, 4.0])),
              corner_modes=int(rng.integers(0, 2)), f_cluster=float(rng.choice([1.0, 0.93])),
              boxsize=float(rng.choice([720.0, 720.0, 90.0, 2000.0])))
    if comp:
        R = int(rng.choice([1, 2, 2, 4])) if plt else int(rng.choice([2, 2, 4]))
        if n // R < 24 or n % R or (n // R) % 8:  # z lines the composite kernels transform: 16 * 2^k * Q, and 8 * Q
            R = 1 if plt else 2
    else:
        R = int(rng.choice([1, 2, 4]))
        if n // R < 32:
            R = 1
    kw["stream_factor"] = R
    kw["store_mode"] = str(rng.choice(["auto", "auto", "auto", "reference", "packed", "fields"]))
    if rng.integers(0, 3) == 0:
        kw["qdensity"] = int(rng.choice([1, 1, 2]))
    if rng.integers(0, 8) == 0:
        h = n // 2
        # (no zero component: a displacement component that is zero but for rounding has no scale to compare against; ky < 0 is
        # never drawn — the half space of zeldovich.cpp:333 — and gives the zero field on both sides, which is a case too)
        sgn = lambda: int(rng.choice([-1, 1]))
        kw.update(qonemode=1, one_mode=[sgn() * int(rng.integers(1, h // 3 + 1)), int(rng.choice([-1, 1, 1, 1])) * int(rng.integers(1, h // 3 + 1)),
                                        sgn() * int(rng.integers(1, h // 3 + 1))])
    # reference arrays (the only store of these options) exist on several ranks for the powers of two only: a composite grid then
    # runs as convolutions on one rank, and says so
    ref_arrays = (kw["store_mode"] == "reference" or (kw["corner_modes"] and kw["k_cutoff"] != 1.0) or (kw["k_cutoff"] < 1.0)
                  or (comp and kw["store_mode"] == "packed"))  # (the composite kernels exist for the field stores only)
    if rng.integers(0, 3) == 0 and n >= 128 and not (comp and (ref_arrays or (plt and kw.get("qdensity") == 1))):
        ng = int(rng.choice([2, 4]))
        if (n // R) % ng == 0 and (n // 2) % (8 * ng) == 0:
            kw["ngpu"] = ng
            kw["pass_groups"] = 1
            # the ZA packings carry two residues per pass (with a density: the six-field store of the composite grids only)
            passes = R if (plt or (kw.get("qdensity") and not comp) or ref_arrays or R < 2) else R // 2
            if rng.integers(0, 2) and passes % ng == 0:
                kw["pass_groups"] = ng
            elif rng.integers(0, 2):
                kw["loopback"] = True  # one group of ranks on the RCCL branch of the exchange (in-process emulation of its calls: the -DZD_TESTING library)
    if rng.integers(0, 6) == 0:  # ZD_Version = 1: mt19937 streams, PPD / NumBlock of them, dealt over the ranks
        if (n // 2) % kw.get("ngpu", 1) == 0 and kw["k_cutoff"] in (1.0, 2.0, 4.0):  # (NumBlock 2 x k_cutoff must divide PPD: parameters.cpp:129-141)
            kw["version"] = 1
    if rng.integers(0, 8) == 0 and "ngpu" not in kw and n <= 192:  # the phi round (composite grids: on the convolution kernels)
        kw.update(f_NL=float(rng.choice([2.0e4, -1.0e4])), n_s=0.96, Omega_M=0.31)
        kw.pop("stream_factor")
    eig = None
    if plt:
        eig = oracle.synthetic_eigenmodes(int(rng.choice([16, 24, 32, 48, 64])), seed=int(rng.integers(0, 100)))
        kw.update(qPLT=1, qPLTrescale=int(rng.integers(0, 2)), PLT_target_z=float(rng.choice([3.0, 9.0])))
    pkw = dict(fix_to_mean=int(rng.integers(0, 2)), Pk_smooth=float(rng.choice([0.0, 0.0, 0.5])))
    fmt = str(rng.choice(["RVdoubleZel", "RVZel", "Zeldovich", "ZelSimple"]))
    return n, fmt, kw, pkw, eig


def test_soak_random_options_vs_oracle(zd, oracle, wmap_path):
    seed = int(os.environ.get("ZD_SOAK_SEED", "1"))
    trials = int(os.environ.get("ZD_SOAK_TRIALS", "60"))
    rng = np.random.default_rng(seed)
    failed = []
    for trial in range(trials):
        n, fmt, kw, pkw, eig = _draw(rng, oracle)
        box = kw["boxsize"]
        ps = zd.PowerSpectrum.from_file(wmap_path, box, **pkw)
        opk = oracle.pk_from_file(wmap_path, box, **pkw)
        if kw.get("f_NL"):
            import ctypes as C
            oracle.lib().zdo_pk_set_primordial(C.byref(opk), kw["n_s"])
        desc = (trial, n, fmt, kw, pkw, None if eig is None else eig.shape[0])
        print("trial", *desc, flush=True)
        try:
            _compare(zd, oracle, ps, opk, n, fmt=fmt, eig=eig, tie_ok=bool(kw.get("qonemode")), **kw)
        except Exception as e:  # keep going: the list of failing combinations is the result
            print("   FAILED:", repr(e)[:300], flush=True)
            failed.append((desc, repr(e)[:300]))
    assert not failed, failed


def test_soak_store_modes_agree_at_mid_sizes(zd, oracle, wmap_path):
    """Sizes the oracle does not reach in seconds (512 ... 1280, powers of two and composite): the same random options through the
    library's different stores — `auto` (packed / field stores, the fused PLT Z stage at 512 and 1024, composite kernels) against
    `reference` (the reference's arrays; composite grids: the convolution kernels) and, where it differs from `auto`, `packed` —
    three sample planes record by record, density_variance, max_disp and its lattice site.  ZD_SOAK_SEED / ZD_SOAK_TRIALS2 /
    ZD_SOAK_SIZES (e.g. "2048,1536,1792": BASELINE C3's size through the fused Z stage against the reference's arrays)."""
    seed = int(os.environ.get("ZD_SOAK_SEED", "1"))
    trials = int(os.environ.get("ZD_SOAK_TRIALS2", "24"))
    rng = np.random.default_rng(1000 + seed)
    failed = []
    for trial in range(trials):
        n = int(rng.choice([int(v) for v in os.environ.get("ZD_SOAK_SIZES", "512,512,1024,1024,768,640,896,1280").split(",")]))
        comp = n & (n - 1) != 0
        plt = bool(rng.integers(0, 2))
        kw = dict(seed=int(rng.integers(1, 2 ** 31 - 1)), k_cutoff=float(rng.choice([1.0, 1.0, 1.0, 1.5, 2.0, 4.0])),
                  corner_modes=int(rng.integers(0, 2)), f_cluster=float(rng.choice([1.0, 0.93])), boxsize=float(rng.choice([720.0, 90.0, 2000.0])))
        eig = None
        if plt:
            eig = oracle.synthetic_eigenmodes(int(rng.choice([32, 48, 64, 128])), seed=int(rng.integers(0, 100)))
            kw.update(qPLT=1, qPLTrescale=int(rng.integers(0, 2)), PLT_target_z=float(rng.choice([3.0, 9.0])))
        if rng.integers(0, 6) == 0:
            h = n // 2
            kw.update(qonemode=1, one_mode=[int(rng.choice([-1, 1])) * int(rng.integers(1, h // 3)), int(rng.integers(1, h // 3)),
                                            int(rng.choice([-1, 1])) * int(rng.integers(1, h // 3))])
        pkw = dict(fix_to_mean=int(rng.integers(0, 2)), Pk_smooth=float(rng.choice([0.0, 0.0, 0.5])))
        ps = zd.PowerSpectrum.from_file(wmap_path, kw["boxsize"], **pkw)
        zs = sorted(int(z) for z in rng.choice(n, 3, replace=False))
        modes = ["reference", "auto"] + (["packed"] if not comp and rng.integers(0, 2) else [])
        desc = (trial, n, kw, pkw, None if eig is None else eig.shape[0], zs, modes)
        print("trial", *desc, flush=True)
        res = {}
        try:
            for m in modes:
                planes = {}

                def keep(z, rec):
                    if z in zs:
                        planes[z] = rec.copy()

                out = zd.generate_planes(zd.make_params(n, icformat="RVdoubleZel", store_mode=m, **kw), ps, keep, eig=eig)
                res[m] = (planes, out)
            ref_planes, ref = res["reference"]
            for m in modes[1:]:
                planes, out = res[m]
                assert out["planes"] == ref["planes"] == n
                for z in zs:
                    a, b = planes[z], ref_planes[z]
                    assert np.array_equal(a["ijk"], b["ijk"])
                    for f in ("d", "v"):
                        scale = np.abs(b[f]).max()
                        assert np.abs(a[f] - b[f]).max() <= 1e-10 * max(scale, 1e-300), (m, z, f, np.abs(a[f] - b[f]).max(), scale)
                assert abs(out["density_variance"] - ref["density_variance"]) <= 1e-10 * ref["density_variance"], m
                md, mr = np.asarray(out["max_disp"]), np.asarray(ref["max_disp"])
                if kw.get("qonemode"):  # a plane wave: +max and -max tie to rounding, either may be the first (see _compare's tie_ok)
                    md, mr = np.abs(md), np.abs(mr)
                assert np.abs(md - mr).max() <= 1e-10 * np.abs(mr).max(), m
        except Exception as e:
            print("   FAILED:", repr(e)[:400], flush=True)
            failed.append((desc, repr(e)[:400]))
    assert not failed, failed
