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
        if n // R < 48 or n % R or (n // R) % 16:  # z lines the composite kernels transform: 16 * 2^k * Q
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
    ref_arrays = kw["store_mode"] == "reference" or (kw["corner_modes"] and kw["k_cutoff"] != 1.0) or (kw["k_cutoff"] < 1.0)
    if rng.integers(0, 4) == 0 and n >= 128 and not (comp and (ref_arrays or (plt and kw.get("qdensity") == 1))):
        ng = int(rng.choice([2, 4]))
        if (n // R) % ng == 0 and (n // 2) % (8 * ng) == 0:
            kw["ngpu"] = ng
            kw["pass_groups"] = 1
            # the ZA packings carry two residues per pass (with a density: the six-field store of the composite grids only)
            passes = R if (plt or (kw.get("qdensity") and not comp) or ref_arrays or R < 2) else R // 2
            if rng.integers(0, 2) and passes % ng == 0:
                kw["pass_groups"] = ng
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
