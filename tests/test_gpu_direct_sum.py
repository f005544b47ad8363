"""GPU records at the FULL BASELINE sizes against the oracle's direct summation over every live mode (VERDICT r3 #1d).

tests/golden/direct_sum.json (made by tests/golden/make_direct_sum.py from oracle/zd_oracle.c: zdo_direct_sum — per-mode draws in
LoadPlane's stream order, eigenmodes, rescale; no FFT, no blocking, no packing; src/zeldovich.cpp:331-503, src/output.cpp:93-141)
holds the displacement and velocity at 8 lattice sites on 4 planes of different residue passes for
  * PPD = 4096 ZA            — the bench workload (BASELINE C4's grid): k_genf / k_zfft_f / k_yfft_f<4096> / k_xfft_seq<4096>
  * PPD = 2048 PLT + rescale — BASELINE C3: k_genf PLTN + k_eig_lines / k_zfft / k_yfft / k_xfft_seq_plt<2048>
  * PPD = 4096 PLT + rescale — k_xfft_seq_plt<4096, 16, true>, R = 16
  * PPD = 8192 ZD_k_cutoff=2 — BASELINE C5's grid on one GPU
  * PPD = 3456 ZA            — the composite-transform kernels (2^7 3^3: k_zfft_fq / k_yfft_fq / k_xfft_seq1_q)
  * PPD = 6912 PLT + rescale — the production Abacus configuration (2^8 3^3, 3.3e11 particles, one GPU, R = 48): the kz-paired PLT
                               generator, the PLT field store and the composite kernels' PLT x stage
This anchors the random-field path AT the headline sizes to the oracle directly, not through a chain of HIP runs.  The CPU suite
re-derives the small cases of the same file from a full oracle run (tests/test_oracle_golden.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, WMAP

pytestmark = pytest.mark.gpu
FIXTURE = json.load(open(os.path.join(GOLDEN, "direct_sum.json")))


@pytest.fixture(scope="module")
def zd():
    import zeldovich_plt_amd.api as api
    api.load_library()
    return api


def records_at(zd, ps, n, sites, eig=None, **kw):
    """RVdoubleZel records of a PPD = n run at the lattice sites [(z, y, x), ...]: only the passes that hold their planes are
    executed (staged API), only the picked records leave the GPU"""
    import torch
    fmt = "RVdoubleZel"
    p = zd.make_params(n, icformat=fmt, **kw)
    if p.stream_factor <= 0:
        free_b, _ = torch.cuda.mem_get_info()
        p.stream_factor = zd.load_library().zd_choose_stream_factor(C.byref(p), 1, int(free_b) - (24 << 30))
        assert p.stream_factor > 0
    plan = zd.Plan(p, ps, eig=eig)
    store = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
    dt = zd.RECORD_DTYPES[fmt]
    step = plan.plane_step
    out = torch.empty(step * n * n * dt.itemsize, dtype=torch.uint8, device="cuda")
    where = {}
    for ps_ in range(plan.passes):
        for lp in range(plan.local_planes):
            where[plan.plane_z(ps_, lp)] = (ps_, lp)
    res = {}
    for z in sorted({s[0] for s in sites}, key=lambda z: where[z]):
        pass_, lp = where[z]
        plan.stage_z(pass_, store.data_ptr())
        plan.stage_y(store.data_ptr())
        first = lp // step * step
        plan.stage_x(pass_, store.data_ptr(), first, step, out.data_ptr())
        torch.cuda.synchronize()
        img = out.view(step, n, n, dt.itemsize)
        for (zz, y, x) in sites:
            if zz == z:
                res[(zz, y, x)] = img[lp - first, y, x].cpu().numpy().view(dt)[0]
    info = dict(R=plan.R, passes=plan.passes, store=plan.store_mode)
    plan.close()
    del store, out
    torch.cuda.empty_cache()
    return res, info


@pytest.mark.parametrize("case", ["ppd256_za", "ppd256_plt", "ppd2048_plt_rescale", "ppd4096_za", "ppd4096_plt_rescale",
                                  "ppd8192_kcut2_za", "ppd96_za", "ppd160_plt", "ppd3456_za", "ppd6912_plt_rescale"])
def test_records_at_sites_equal_the_direct_sum_over_all_modes(zd, oracle, case):
    c = FIXTURE[case]
    n = c["ppd"]
    assert c["seed"] == 12346 and c["boxsize"] == 720.0
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    eig = oracle.synthetic_eigenmodes(c["eig_ppd"]) if c["eig_ppd"] else None
    sites = [tuple(s) for s in c["sites"]]
    got, info = records_at(zd, ps, n, sites, eig=eig, **c["params"])
    want = np.array(c["values"])
    scale = np.abs(want[:, :3]).max()
    assert scale > 1e-3  # Mpc/h: a real displacement field
    worst = 0.0
    for s, w in zip(sites, want):
        r = got[s]
        assert tuple(int(v) for v in r["ijk"]) == s
        d = np.array(r["d"])[::-1]  # records hold (qz, qy, qx)
        v = np.array(r["v"])[::-1]
        worst = max(worst, np.abs(d - w[:3]).max() / scale, np.abs(v - w[3:6]).max() / scale)
    print(case, info, "max |GPU - direct sum| / max|q| =", worst)
    assert worst < 1e-10
