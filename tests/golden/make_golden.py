#!/usr/bin/env python3
"""Generates tests/golden/pcg_kat.json and tests/golden/spline_kat.json from the REFERENCE's own
object code (oracle/_ref/libzd_ref.so: include/pcg-rng/pcg_random.hpp and include/spline_function.h
compiled where they lie under /root/reference).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden.py

The JSON files are data (inputs + expected outputs); no reference source travels.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import zdo  # noqa: E402

R = zdo.ref()
u64 = C.c_uint64


def seed_state(seed):
    hi, lo = u64(), u64()
    R.ref_pcg_seed(u64(seed & (2 ** 64 - 1)), C.byref(hi), C.byref(lo))
    return hi.value, lo.value


def draws(state, n):
    hi, lo = u64(state[0]), u64(state[1])
    out = (u64 * n)()
    R.ref_pcg_draw(C.byref(hi), C.byref(lo), n, out)
    return [int(x) for x in out], (hi.value, lo.value)


def advance(state, delta):
    hi, lo = u64(state[0]), u64(state[1])
    R.ref_pcg_advance(C.byref(hi), C.byref(lo), u64(delta >> 64), u64(delta & (2 ** 64 - 1)))
    return hi.value, lo.value


def main():
    rng = np.random.default_rng(20261003)
    kat = {"source": "reference pcg64 (include/pcg-rng/pcg_random.hpp) via oracle/_ref", "seeds": [], "modes": []}
    for seed in [12346, 0, 1, 2 ** 31 - 1, -5, 987654321]:
        s0 = seed_state(seed)
        d, s1 = draws(s0, 6)
        plane = advance(s0, 2 * 65536 * 65536)
        big = advance(s0, (1 << 70) + 12345)
        kat["seeds"].append({"seed": seed, "state0": [hex(s0[0]), hex(s0[1])], "draws": [hex(x) for x in d],
                             "after_draws": [hex(s1[0]), hex(s1[1])],
                             "after_plane_advance": [hex(plane[0]), hex(plane[1])],
                             "after_2p70_12345": [hex(big[0]), hex(big[1])]})
    # per-mode counters for seed 12346: the reference stream advanced to the
    # mode's slot, then two draws
    s0 = seed_state(12346)
    modes = [(3, 5, 7), (-3, 5, 7), (3, 5, -7), (-9, 1, -2), (1, 0, 2), (0, 2, 0)]
    for _ in range(58):
        modes.append((int(rng.integers(-2047, 2048)), int(rng.integers(0, 2048)), int(rng.integers(-2047, 2048))))
    for kx, ky, kz in modes:
        c = 2 * ((ky * 65536 + (kz & 65535)) * 65536 + (kx & 65535))
        d, _ = draws(advance(s0, c), 2)
        kat["modes"].append({"k": [kx, ky, kz], "counter": c, "r": [hex(d[0]), hex(d[1])]})
    json.dump(kat, open(os.path.join(HERE, "pcg_kat.json"), "w"), indent=1)

    # spline: the wmap1new table nodes (ln k, ln P) -> SplineFunction::val at probe points
    tab = np.loadtxt(os.path.join(HERE, "wmap1new.pow"))
    x, y = np.log(tab[:, 0]), np.log(tab[:, 1])
    probes = np.concatenate([np.linspace(x[0] - 0.5, x[-1] + 1.5, 97), x[::17], rng.uniform(x[0], x[-1], 40)])
    out = np.zeros_like(probes)
    R.ref_spline_val(len(x), x.ctypes.data, y.ctypes.data, len(probes), probes.ctypes.data, out.ctypes.data)
    # also an unsorted node order (exercises sort_arrays)
    perm = rng.permutation(len(x))
    xs, ys = np.ascontiguousarray(x[perm]), np.ascontiguousarray(y[perm])
    out2 = np.zeros_like(probes)
    R.ref_spline_val(len(xs), xs.ctypes.data, ys.ctypes.data, len(probes), probes.ctypes.data, out2.ctypes.data)
    json.dump({"source": "reference SplineFunction (include/spline_function.h) via oracle/_ref, nodes = ln of wmap1new.pow",
               "probes": [float.hex(float(v)) for v in probes], "val": [float.hex(float(v)) for v in out],
               "perm": [int(i) for i in perm], "val_permuted_nodes": [float.hex(float(v)) for v in out2]},
              open(os.path.join(HERE, "spline_kat.json"), "w"), indent=1)
    print("wrote pcg_kat.json, spline_kat.json")

    # BlockArray: StoreBlock for every block, then LoadBlock for every block, through the reference's own object code
    # (src/block_array.cpp:387-414,466-504, RAM mode).  Input = oracle.zdo.blockarray_input (coordinates as values).
    import hashlib
    cases = []
    for ppd, nb, na in [(8, 2, 2), (8, 4, 4), (8, 2, 1), (16, 2, 4), (16, 4, 2), (16, 8, 4), (32, 4, 2)]:
        arr, slabs = zdo.blockarray_roundtrip(R.ref_blockarray_roundtrip, ppd, nb, na)
        case = {"ppd": ppd, "numblock": nb, "narray": na, "fill": -7.5,
                "arr_sha256": hashlib.sha256(arr.tobytes()).hexdigest(),
                "slabs_sha256": hashlib.sha256(slabs.tobytes()).hexdigest()}
        if ppd == 8:  # small cases in full: real parts (the imaginary part of every input element is -re - 0.25)
            case["arr_re"] = [int(v) for v in arr[:, 0]]
            case["slabs_re"] = [float(v) for v in slabs[:, 0]]
        cases.append(case)
    json.dump({"source": "reference BlockArray (src/block_array.cpp + src/STimer.cc, RAM mode) via oracle/_ref; input = "
                         "oracle.zdo.blockarray_input: element (y, a, z, x) of the z-stage slabs holds re = 1 + linear "
                         "index, im = -re - 0.25; arr = BlockArray image after all StoreBlock calls, slabs = xy-stage "
                         "slabs [zblock][zres][a][y][x] after all LoadBlock calls (pre-filled with `fill`)",
               "cases": cases}, open(os.path.join(HERE, "blockarray_kat.json"), "w"), indent=1)
    print("wrote blockarray_kat.json")


if __name__ == "__main__":
    main()
