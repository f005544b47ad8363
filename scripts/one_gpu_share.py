"""What ONE GPU of an N-GPU job does per step when the library's policy gives every GPU its own pass group (no exchange):
its share of the residue passes, timed on this one GPU.  A measured entry for DESIGN.md §5's prediction table.
    python scripts/one_gpu_share.py --ppd 4096 --of 8 [--plt 1]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ppd", type=int, default=4096)
    ap.add_argument("--of", type=int, default=8, help="GPUs of the job")
    ap.add_argument("--plt", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    import zeldovich_plt_amd.api as zd
    from zeldovich_plt_amd.parallel import HipEngine, SlabPipeline
    N = args.ppd
    L = zd.load_library()
    ps = zd.PowerSpectrum.from_file(bench.WMAP, 720.0)
    eig = bench.synthetic_eigenmodes(128) if args.plt else None
    kw = dict(numblock=64 if N >= 4096 else 4, icformat="RVZel", profile=1)
    if args.plt:
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
    p = zd.make_params(N, **kw)
    budget = int(torch.cuda.mem_get_info()[0]) - (16 << 30)
    g, R = ctypes.c_int32(), ctypes.c_int32()
    assert L.zd_choose_pass_groups(ctypes.byref(p), args.of, budget, ctypes.byref(g), ctypes.byref(R)) == 0
    if g.value != args.of:
        raise SystemExit("the policy gives %d group(s) for %d GPUs: the ranks exchange, not measurable on one GPU" % (g.value, args.of))
    p.stream_factor = R.value
    plan = zd.Plan(p, ps, eig=eig, rank=0, nranks=1)
    pipe = SlabPipeline(HipEngine(plan, N), N, device="cuda")
    out = {"ppd": N, "plt": args.plt, "gpus": args.of, "stream_factor": R.value, "passes": plan.passes,
           "passes_per_gpu": plan.passes // args.of, "ranks": []}
    for rank in (0, args.of - 1):
        pipe.run(pass_first=rank, pass_step=args.of)
        torch.cuda.synchronize()
        plan.stats()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.run(pass_first=rank, pass_step=args.of)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        st = plan.stats()
        out["ranks"].append({"rank": rank, "s_per_step": dt,
                             "kernel_ms": {k: v / args.steps for k, v in st["kernel_ms"].items()}})
    one = out["ranks"]
    out["predicted_s_per_step"] = max(r["s_per_step"] for r in one)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
