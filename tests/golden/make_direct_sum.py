"""Generates tests/golden/direct_sum.json: displacement / velocity at a few lattice sites of the FULL-SIZE BASELINE
workloads, by the oracle's direct summation over every live mode (oracle/zd_oracle.c: zdo_direct_sum — per-mode draws in
LoadPlane's stream order, no FFT, no blocking, no packing).  The GPU tests compare the records of the real runs at those
sites (tests/test_gpu_direct_sum.py); a CPU test re-derives the small case from the oracle (tests/test_oracle_golden.py).

    python tests/golden/make_direct_sum.py [case ...]      # minutes to an hour of CPU per full-size case (OpenMP)

Inputs are the committed wmap1new.pow and oracle.zdo.synthetic_eigenmodes(128) (deterministic)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import zdo  # noqa: E402

WMAP = os.path.join(HERE, "wmap1new.pow")
OUT = os.path.join(HERE, "direct_sum.json")
PLT = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)


def sites_for(n):
    # 8 sites on 4 planes (different residue passes), away from any symmetry point
    zs = [5, n // 2 + 3, n - 2, n // 4 + 6]
    out = []
    for i, z in enumerate(zs):
        out.append((z, (37 * (i + 1) + n // 3) % n, (101 * (i + 2) + n // 5) % n))
        out.append((z, (n - 11 * (i + 1)) % n, (n // 2 + 17 * (i + 1)) % n))
    return out


CASES = {
    "ppd256_za": dict(n=256, kw={}),                       # re-derived on CPU by the test suite
    "ppd256_plt": dict(n=256, kw=PLT, eig=128),
    "ppd2048_plt_rescale": dict(n=2048, kw=PLT, eig=128),  # BASELINE C3
    "ppd4096_za": dict(n=4096, kw={}),                     # the bench workload / BASELINE C4's grid
    "ppd4096_plt_rescale": dict(n=4096, kw=PLT, eig=128),
    "ppd8192_kcut2_za": dict(n=8192, kw=dict(k_cutoff=2.0)),  # BASELINE C5's grid
    "ppd96_za": dict(n=96, kw={}),                         # composite grids, small: re-derived on CPU against the oracle's plain-DFT path
    "ppd160_plt": dict(n=160, kw=PLT, eig=32),
    "ppd3456_za": dict(n=3456, kw={}),                     # composite-transform kernels (2^7 3^3)
    "ppd6912_plt_rescale": dict(n=6912, kw=PLT, eig=128),  # the production Abacus configuration: 2^8 3^3 with PLT + rescale, one GPU
}


def main():
    want = sys.argv[1:] or list(CASES)
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    pk = zdo.pk_from_file(WMAP, 720.0)
    for name in want:
        c = CASES[name]
        n = c["n"]
        p = zdo.make_params(n, numblock=2, **c["kw"])
        eig = zdo.synthetic_eigenmodes(c["eig"]) if c.get("eig") else None
        sites = sites_for(n)
        t = time.time()
        ds = zdo.direct_sum(p, pk, sites, eig=eig)
        dt = time.time() - t
        data[name] = dict(ppd=n, boxsize=720.0, seed=12346, params=c["kw"], eig_ppd=c.get("eig", 0), sites=sites,
                          fields=["qx", "qy", "qz", "vx", "vy", "vz", "density"], values=[[float(v) for v in row] for row in ds],
                          cpu_seconds=round(dt, 1))
        print(name, "%.1f s" % dt, flush=True)
        json.dump(data, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
