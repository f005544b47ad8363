#!/usr/bin/env python3
"""bench.py — grid->displacements throughput of the MI355X path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--ppd P] [--plt 0|1] [--format RVZel] [--stream R]

One "step" = one complete pass of the hot path over the workload: mode generation, Hermitian
packing + z FFT, (all-to-all between ranks), y FFT, x FFT + particle epilogue for all ppd^3
particles.  Inputs (P(k) spline, eigenmode table, RNG jump tables) are resident in HBM before
the timed region; finished planes are produced into an HBM ring and dropped (no PCIe in `value`).
For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL);
the fixed workload is sharded over the ranks ("strong" scaling) with ONE exchange per residue
pass between the Z and XY stages, done inside the library in plane groups over RCCL and overlapped with the XY stages.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with `roofline`
(dominant kernel, hipEvent-timed on the launch stream inside the timed region) and
`cpu_baseline` (the oracle — CPU port of the reference — timed on the host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
WMAP = os.path.join(ROOT, "tests", "golden", "wmap1new.pow")


def synthetic_eigenmodes(ppd_e=128, seed=7, amp=0.03, singular=False):
    """Synthetic PLT eigenmode table in the reference's file layout (src/zeldovich.cpp:796-797,815,
    155-159): float64 [ppd_e][ppd_e][ppd_e/2+1][4] = (e_x,e_y,e_z,lambda), FFT order, |e|=1.
    eigmodes128 is absent from the reference mount, so PLT paths are exercised with this table:
    e = normalised(k_hat + amp min((k/k_Ny)^2, 1.5) M k_hat) with a fixed random 3x3 matrix M (entries in
    [-1, 1]) — within ~5 degrees of k_hat like real PLT eigenvectors, so k^2/(k.e) stays O(k) and the
    displacements O(1) — and lambda = 1 - 0.2 (k/k_Ny)^2.
    singular=True is the round-1 recipe (perturbation up to ~1.3 |k_hat|): e can come out almost
    perpendicular to k, k^2/(k.e) blows up at a handful of modes; kept as an edge-case input."""
    h = ppd_e // 2 + 1
    idx = np.arange(ppd_e)
    kfull = np.where(idx > ppd_e // 2, idx - ppd_e, idx).astype(np.float64)
    kx, ky, kz = kfull[:, None, None], kfull[None, :, None], np.arange(h, dtype=np.float64)[None, None, :]
    kn = ppd_e / 2.0
    k2 = kx * kx + ky * ky + kz * kz
    kk = np.sqrt(np.where(k2 > 0, k2, 1.0))
    q2 = k2 / (kn * kn)
    c = np.random.RandomState(seed).uniform(-1, 1, size=(3, 3))
    if singular:
        pert = [0.15 * q2 * (c[i, 0] * kx + c[i, 1] * ky + c[i, 2] * kz) / kn for i in range(3)]
    else:
        pert = [amp * np.minimum(q2, 1.5) * (c[i, 0] * kx + c[i, 1] * ky + c[i, 2] * kz) / kk for i in range(3)]
    e = [kx / kk + pert[0], ky / kk + pert[1], kz / kk + pert[2]]
    mag = np.sqrt(e[0] ** 2 + e[1] ** 2 + e[2] ** 2)
    mag = np.where(mag > 0, mag, 1.0)
    out = np.empty((ppd_e, ppd_e, h, 4), dtype=np.float64)
    for i in range(3):
        out[..., i] = e[i] / mag
    out[..., 3] = 1.0 - 0.2 * q2
    out[0, 0, 0, :3] = 0.0
    return np.ascontiguousarray(out)


def source_sha():
    """sha-256 over the native sources (= `make -C zeldovich_plt_amd/csrc srcsha`, tests/conftest.py source_sha)"""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "zeldovich_plt_amd", "csrc")
    names = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))
                   + [os.path.join(ROOT, "include", "zeldovich_hip.h")])
    h = hashlib.sha256()
    for n in names:
        h.update(open(n, "rb").read())
    return h.hexdigest()


def host_memory_available():
    """bytes this process may still allocate on the host: MemAvailable of /proc/meminfo, capped by the cgroup limit when there is one"""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                room = int(v) - int(open(cur).read().strip())
                avail = room if avail is None else min(avail, room)
        except (OSError, ValueError):
            pass
    return avail if avail is not None else 8 << 30


def host_threads():
    """threads the host really gives this process: CPUs of its affinity mask, capped by the cgroup CPU quota (a 1-GPU box shows
    every core of the machine in os.cpu_count() but may schedule only its share)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline_sample(ppd, plt, recsize, cores, avail, rate):
    """(n, numblock) of the bounded sample: the largest power-of-two grid <= min(ppd, 1024) that (i) fits half of the host's free memory
    — the oracle holds the reference's BlockArray (n^3 a Complx), its two generation slabs + one XY slab (3 n^3 a / NumBlock) and the
    records — and (ii) is at most ~35 s of work at `rate` particles/s (measured by the caller on a PPD=256 probe run on these threads).
    NumBlock = the largest that still gives every OpenMP loop of ZeldovichZ / ZeldovichXY (src/zeldovich.cpp:572-577,653-658:
    over the n / NumBlock planes of a block) two iterations per thread, at least 2."""
    a = 4 if plt else 2
    n = 1024
    while n > 64:
        nb = 2
        while nb * 2 <= n // 2 and n // (nb * 2) >= 2 * cores:
            nb *= 2
        need = n ** 3 * (16.0 * a * (1 + 3.0 / nb) + recsize) * 1.1
        secs = n ** 3 / rate
        if n <= ppd and need <= 0.5 * avail and secs <= 35:
            return n, nb
        n //= 2
    return 64, 2


def cpu_baseline(ppd, plt, fmt, eig):
    """oracle (CPU port of the reference path) on the host cores, bounded sample of the workload"""
    from oracle import zdo
    zdo.build()
    cores = host_threads()
    avail = host_memory_available()
    pk = zdo.pk_from_file(WMAP, 720.0)
    kwp = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0) if plt else {}

    def timed(n, nb):
        p = zdo.make_params(n, numblock=nb, icformat=fmt, nthreads=cores, **kwp)
        st = zdo.run(p, pk, eig=eig if plt else None, eig_ppd=eig.shape[0] if plt else 0)["stats"]
        return st, st.t_stage1 + st.t_store + st.t_load + st.t_fft2d + st.t_write

    t0 = time.time()
    npr = min(256, ppd)
    _, tp = timed(npr, 2)  # probe: the rate of these threads on a grid that takes about a second (it does not fill many threads yet)
    n, nb = cpu_baseline_sample(ppd, plt, zdo.RECORD_DTYPES[fmt].itemsize, cores, avail, npr ** 3 / tp)
    st, t = timed(n, nb)
    fft = zdo.fft_backend()  # "fftw3" when the host has libfftw3.so.3 (the reference's transform), else the oracle's own
    return {"value": n ** 3 / t, "unit": "particles/s", "cores": cores, "threads": cores, "host_cpus_listed": os.cpu_count(), "ppd": n,
            "numblock": nb, "kind": "port",
            "fft": fft, "host_memory_available_GB": avail / 1e9, "timed_seconds": t, "wall_seconds": time.time() - t0,
            "phase_seconds": {"LoadPlane+zFFT": st.t_stage1, "StoreBlock": st.t_store, "LoadBlock": st.t_load,
                              "FFT2D": st.t_fft2d, "WriteParticlesSlab": st.t_write},
            "sample": "PPD=%d NumBlock=%d %s %s, full grid->displacements (ZeldovichZ+ZeldovichXY timers), OpenMP on %d threads "
                      "(%d planes per block: every parallel loop has >= %d iterations per thread), %s; wall %.1fs" % (
                          n, nb, "PLT+rescale" if plt else "ZA", fmt, cores, n // nb, max(1, n // nb // cores),
                          "FFTW3 (dlopen)" if fft == "fftw3" else "radix-2 CPU FFT (no libfftw3.so.3 on this host)",
                          time.time() - t0)}


def arm_last_words(make_line):
    """C-level handlers for SIGSEGV / SIGABRT / SIGBUS / SIGTERM that write make_line(signal) to stdout and end the process with
    status 128 + signal: the line survives, and the launcher still sees a failed rank (a faulted run must not read as rc = 0).
    A Python-level handler would not run while the main thread sits inside a C call (a collective, a kernel wait); this one is a
    ctypes callback installed with signal(2), entered on the interrupted thread, which takes the GIL the call had released.  The
    lines are serialised when the handlers are armed, so that the handler itself only writes bytes and exits.
    Returns (callback to keep alive, disarm())."""
    import ctypes
    import signal
    libc = ctypes.CDLL(None)
    libc.signal.restype = ctypes.c_void_p
    libc.signal.argtypes = [ctypes.c_int, ctypes.c_void_p]
    fatal = (signal.SIGSEGV, signal.SIGABRT, signal.SIGBUS, signal.SIGTERM)
    lines = {int(sg): (make_line(int(sg)) + "\n").encode() for sg in fatal}

    def _handler(sig):
        try:
            os.write(1, lines.get(sig, b""))
        finally:
            os._exit(128 + sig)

    cb = ctypes.CFUNCTYPE(None, ctypes.c_int)(_handler)
    for sg in fatal:
        libc.signal(int(sg), ctypes.cast(cb, ctypes.c_void_p))

    def disarm():
        for sg in fatal:
            libc.signal(int(sg), None)  # SIG_DFL

    return cb, disarm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # default workload = the configuration BASELINE.json's metric is quoted on: PPD=4096 (configs[3], ZA, NumBlock=64);
    # it runs on ONE GPU through z-residue streaming (R=8).  --ppd 2048 --plt 1 is configs[2].
    ap.add_argument("--ppd", type=int, default=int(os.environ.get("ZD_BENCH_PPD", "4096")))
    ap.add_argument("--plt", type=int, default=int(os.environ.get("ZD_BENCH_PLT", "0")))
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra untimed pass that times every kernel alone")
    ap.add_argument("--format", default="RVZel")
    ap.add_argument("--stream", type=int, default=0, help="z-residue stream factor R (0 = auto)")
    ap.add_argument("--groups", type=int, default=0,
                    help="N > 1: independent groups of GPUs, residue passes dealt round-robin over them (0 = the library's choice: "
                         "one GPU per group while the passes, after at most one doubling of the stream factor, deal out over the N GPUs; else "
                         "one group with the all-to-all exchange)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--store-mode", default=os.environ.get("ZD_BENCH_STORE_MODE", "auto") or "auto",
                    help="zd_params.store_mode for A/B runs (auto | reference | packed | fields): PLT with `packed` keeps the two-kernel "
                         "Z stage where `auto` takes the fused generator + z FFT")
    ap.add_argument("--two-stores", action="store_true",
                    help="one GPU: a second block store, the Z stage of pass p + 1 issued beside the y / x stages of pass p "
                         "(zd_plan_run_passes); use with --stream 2R so that two stores fit (A/B measurement, DESIGN §8)")
    ap.add_argument("--dist", action="store_true",
                    help="take the N > 1 code path (torch.distributed RCCL group, zd.Comm id broadcast, per-rank gathers) whatever "
                         "the world size: rehearses it on one GPU under torch.distributed.run --nproc-per-node 1")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL between processes needs on this driver; normally exported already)
    import torch
    import zeldovich_plt_amd.api as zd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.dist  # the N > 1 code path
    if multi:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    N, plt, fmt = args.ppd, bool(args.plt), args.format
    zd.load_library()
    ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
    eig = synthetic_eigenmodes(128) if plt else None
    kw = dict(numblock=64 if N >= 4096 else 4, icformat=fmt, profile=1,
              store_mode=int(args.store_mode) if args.store_mode.isdigit() else args.store_mode)
    if plt:
        kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
    p = zd.make_params(N, **kw)
    narray = 4 if plt else 2
    recsize = zd.RECORD_DTYPES[fmt].itemsize

    free_b, total_b = torch.cuda.mem_get_info()
    import ctypes
    from zeldovich_plt_amd.parallel import HipEngine, SlabPipeline, split_ranks
    budget = int(free_b) - (16 << 30)  # tables, ~3 GB of folded-input slabs, 8 GB record ring, runtime
    if multi:  # every rank must arrive at the same split and R: use the smallest budget of the job
        t = torch.tensor([budget], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        budget = int(t.item())
    # N > 1: the links are probed before anything is decided — a communicator over ALL ranks, 128 MB to and from every peer at once,
    # three repetitions (zd_comm_probe); the slowest rank's figure prices the all-to-all in the library's choice between the two
    # splits (zd_choose_pass_groups_measured) instead of an assumed link rate
    link = {"GBps_per_peer": 0.0, "bytes_per_peer": 128 << 20, "reps": 3}
    if multi:
        def world_id(raw):
            mine = torch.tensor(list(raw), dtype=torch.uint8, device="cuda")
            dist.broadcast(mine, src=0)
            return bytes(mine.cpu().tolist())
        try:
            pc = zd.Comm(rank, world, world_id)
            r = torch.tensor([pc.probe(link["bytes_per_peer"], link["reps"])], dtype=torch.float64, device="cuda")
            pc.close()
            dist.all_reduce(r, op=dist.ReduceOp.MIN)
            link["GBps_per_peer"] = float(r.item())
        except Exception as e:  # (every rank fails alike: a communicator that does not come up fails on all of them)
            link["error"] = repr(e)
    choice_est = {}

    def run_split(groups_req):
        """one measurement of the workload with the GPUs split into `groups_req` pass groups (0 = the library's choice): plan,
        communicator, buffers, W warm-up + K timed steps between fences, per-rank spans.  The caller closes what it returns."""
        # How the GPUs share the job (the library's policy, zd_choose_pass_groups): `groups` independent groups of gsz ranks;
        # group j runs the residue passes j, j + groups, ...; inside a group the rows / planes are sharded with one exchange per pass
        p.stream_factor = args.stream
        p.pass_groups = groups_req
        g_, R_ = ctypes.c_int32(), ctypes.c_int32()
        est = (ctypes.c_double * 2)()
        if zd.load_library().zd_choose_pass_groups_measured(ctypes.byref(p), world, budget, link["GBps_per_peer"], ctypes.byref(g_),
                                                            ctypes.byref(R_), est):
            return None
        if groups_req == 0 and est[0] > 0:
            choice_est.update(pass_groups_s=est[0], all_to_all_s=est[1])
        groups, R = g_.value, R_.value
        p.stream_factor = R
        grp_id, grank, gsz = split_ranks(rank, world, groups)
        plan = zd.Plan(p, ps, eig=eig, rank=grank, nranks=gsz)
        assert plan.passes % groups == 0
        comm = None
        if multi and (gsz > 1 or args.dist):
            # the library's own RCCL communicator, one per group; torch.distributed only carries the 128-byte ids (rank 0 of every
            # group makes one) and the timing fences.  zd_comm_create ends with a 1 MB grouped send / receive between neighbours:
            # a mis-wired communicator fails here, in milliseconds, not inside pass 0
            def exchange_id(raw):
                mine = torch.tensor(list(raw), dtype=torch.uint8, device="cuda")
                allr = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine)
                return bytes(allr[grp_id * gsz].cpu().tolist())
            comm = zd.Comm(grank, gsz, exchange_id)
        pipe = SlabPipeline(HipEngine(plan, N, comm=comm), N, world=gsz, dist=dist, device="cuda")
        pipelined = False
        if comm is not None and gsz > 1 and plan.passes // groups >= 2:
            free_now = torch.tensor([torch.cuda.mem_get_info()[0]], dtype=torch.int64, device="cuda")
            dist.all_reduce(free_now, op=dist.ReduceOp.MIN)  # every rank of the job takes the same decision
            if int(free_now.item()) > plan.exchange_bytes + (12 << 30):
                pipe.alloc_second_store()  # passes pipelined: Z stage of pass p + 1 beside the exchange of pass p
                pipelined = True
        if args.two_stores and gsz == 1 and plan.passes // groups >= 2:
            pipe.alloc_second_store()  # one GPU: Z stage of pass p + 1 beside the XY stages of pass p
            pipelined = True

        def step():
            # per residue pass, inside the library (zd_plan_run_pass): Z stage -> exchange in plane groups over RCCL/xGMI,
            # overlapped with -> y FFT -> x FFT + epilogue of the previous group
            pipe.run(pass_first=grp_id, pass_step=groups)

        for _ in range(args.warmup):
            step()
        fence()
        plan.stats()  # drop warm-up kernel timers
        if comm is not None:
            comm.traffic(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        st = plan.stats()
        per_rank = None
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            # what every rank did in the timed region, so that a scaling curve can be read: Z stage, time the compute stream
            # stood waiting for exchanged planes, XY stages (hipEvent spans on the launch stream), bytes sent to peers
            mine = torch.tensor([dt, st["kernel_ms"]["z_stage"], st["kernel_ms"]["exchange_wait"], st["kernel_ms"]["k_yfft"],
                                 st["kernel_ms"]["k_xfft"], float(st["bytes_sent"])], dtype=torch.float64, device="cuda")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            dt = float(t.item())
            per_rank = [{"rank": i, "wall_ms_per_step": float(v[0]) / args.steps * 1e3, "z_stage_ms": float(v[1]) / args.steps,
                         "exchange_wait_ms": float(v[2]) / args.steps, "y_ms": float(v[3]) / args.steps,
                         "x_ms": float(v[4]) / args.steps, "GB_sent_per_step": float(v[5]) / args.steps / 1e9,
                         "send_GBps_while_waiting_or_computing": (float(v[5]) / 1e9) / max(float(v[0]), 1e-9)}
                        for i, v in enumerate(allr)]
            if gsz > 1:  # ranks that exchange: RCCL must have been on the data path of EVERY rank
                assert all(r["GB_sent_per_step"] > 0 for r in per_rank), per_rank
        return dict(groups=groups, R=R, gsz=gsz, grp_id=grp_id, plan=plan, comm=comm, pipe=pipe, pipelined=pipelined, dt=dt, st=st,
                    per_rank=per_rank)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def mode_summary(m):
        """what the JSON line says about one split of the GPUs"""
        pl_ = m["plan"]
        d = {"groups": m["groups"], "ranks_per_group": m["gsz"], "stream_factor": m["R"], "passes": pl_.passes,
             "passes_per_gpu": pl_.passes // m["groups"], "s_per_step": m["dt"] / args.steps,
             "particles_per_s": float(N) ** 3 * args.steps / m["dt"], "pipelined_over_two_send_stores": m["pipelined"],
             "block_store_GB_per_rank_and_pass": pl_.exchange_bytes / 1e9}
        if m["per_rank"] is not None:
            d["per_rank"] = m["per_rank"]
            d["GB_sent_per_step"] = sum(r["GB_sent_per_step"] for r in m["per_rank"])
            d["max_exchange_wait_ms"] = max(r["exchange_wait_ms"] for r in m["per_rank"])
        return d

    def close_mode(m):
        m["plan"].close()
        if m["comm"] is not None:
            m["comm"].close()
        m["pipe"] = m["plan"] = m["comm"] = None
        torch.cuda.empty_cache()

    t_start = time.perf_counter()
    main_mode = run_split(args.groups)
    if main_mode is None:
        raise SystemExit("PPD=%d does not fit %d GPU(s)" % (N, world))
    groups, R, gsz, plan, comm, pipe, pipelined = (main_mode[k] for k in ("groups", "R", "gsz", "plan", "comm", "pipe", "pipelined"))
    dt, st, per_rank = main_mode["dt"], main_mode["st"], main_mode["per_rank"]

    # one extra UNTIMED pass with the two-stream overlap of the Z stage switched off: every kernel alone on the chip
    # (in the timed region k_gen and k_zfft share it, so their hipEvent spans there include each other)
    iso = None
    if world == 1 and not args.no_isolated:
        p.serial_z = 1
        plan_iso = zd.Plan(p, ps, eig=eig, rank=0, nranks=1)
        p.serial_z = 0
        pipe.e = HipEngine(plan_iso, N)
        pipe.native = True
        pipe.run()
        torch.cuda.synchronize()
        iso = plan_iso.stats()
        plan_iso.close()

    if rank == 0:
        particles = float(N) ** 3
        value = particles * args.steps / dt
        # SURVEY §8d algorithmic bytes per particle, split over the three units of the path (64*a + S_out in all; a = the
        # REFERENCE's array count).  The Z stage is ONE unit: generator and z FFT overlap on two streams, their own
        # hipEvent spans include each other; `z_stage` is first generator launch .. last z FFT of a pass.
        alg = {"z_stage": 16.0 * narray, "k_yfft": 32.0 * narray, "k_xfft": 16.0 * narray + recsize}
        kms, kl = st["kernel_ms"], st["kernel_launches"]
        # bytes this implementation moves BY CONSTRUCTION per step (every buffer touched once; what rocprof's PMC
        # counters should approach): store = block store of all passes, inter = what the y stage hands to the x stage
        store_b = float(plan.exchange_bytes) * plan.passes
        fields = plan.store_mode == "fields"
        inter_b = 1.5 * particles * 16 if fields else store_b
        fused_z = kl["k_zfft"] == 0 and kl["k_gen"] > 0  # PLT on one rank at z lines of 1024: generator + z FFT are one kernel
        design = {"z_stage": (1.0 if fused_z else 3.0) * store_b,  # (folded inputs written + read,) store written
                  "k_yfft": (store_b + inter_b) if fields else 2.0 * store_b,
                  "k_xfft": inter_b + recsize * particles}
        # PMC bytes per launch come from a rocprofv3 --pmc run of THIS command (scripts/gpu_profile.sh), committed as
        # profiles/traffic_latest.json together with the sha-256 of the native sources it profiled; a figure measured on other
        # kernels than the ones in this tree is not reported (traffic: null + traffic_stale)
        traffic_file, traffic_stale = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("workload") == "PPD=%d plt=%d" % (N, int(plt)) and tj.get("store_arrays") == plan.narray \
                        and tj.get("passes") == plan.passes:
                    if tj.get("source_sha") == source_sha():
                        traffic_file = tj
                    else:
                        traffic_stale = ("profiles/traffic_latest.json was measured on sources %s, this tree is %s: re-run "
                                         "scripts/gpu_profile.sh" % (str(tj.get("source_sha"))[:12], source_sha()[:12]))
            except Exception:
                traffic_file = None

        def unit(k, ms, launches):
            if ms <= 0 or launches <= 0:
                return None
            per_launch = alg[k] * particles * args.steps / world / launches
            d = {"ms_per_step": ms / args.steps, "launches_per_step": launches / args.steps, "avg_launch_ms": ms / launches,
                 "alg_bytes_per_particle": alg[k], "alg_bytes_per_launch": per_launch,
                 "alg_GBps": per_launch / (ms / launches * 1e-3) / 1e9,
                 "design_bytes_per_step": design[k] / world,
                 "design_GBps": design[k] / world * args.steps / (ms * 1e-3) / 1e9}
            if traffic_file:
                d["pmc_bytes_per_launch"] = traffic_file.get("bytes_per_launch", {}).get(k)
            return d

        per_kernel = {k: unit(k, kms[k], kl[k]) for k in alg}
        for k in ("k_gen", "k_zfft"):  # members of the Z stage, spans overlap each other in the timed region
            per_kernel[k] = {"ms_per_step": kms[k] / args.steps, "launches_per_step": kl[k] / args.steps,
                             "note": "overlapped span inside z_stage"}
        per_kernel["k_gen"]["bound"] = ("fp64/int VALU: 2 pcg64 steps + Box-Muller + P(k) per mode, regenerated for each "
                                        "of the %d passes" % plan.passes)
        if fused_z:
            per_kernel["k_gen"]["note"] = ("k_genz_plt: generator + z FFT + Hermitian stores in one kernel (the folded inputs stay on the "
                                           "CU); = the whole z_stage")
        dom = max(alg, key=lambda k: kms[k])
        du = per_kernel[dom]
        dom_kernel, dom_launches = dom, kl[dom]
        if dom == "z_stage" and kms["k_zfft"] > 0 and kl["k_zfft"] > 0:
            # The Z stage is two kernels on two streams; rocprof has no row for the stage.  Its roofline entry is quoted on the member
            # that paces it — the z FFT, whose launches span 97 % of the stage while it runs beside the generator — with the stage's
            # algorithmic bytes per z-FFT launch over the z FFT's hipEvent launch time (= rocprof's AverageNs of k_zfft_f / k_zfft).
            zl = kl["k_zfft"]
            per_launch = alg["z_stage"] * particles * args.steps / zl
            du = {"alg_GBps": per_launch / (kms["k_zfft"] / zl * 1e-3) / 1e9, "avg_launch_ms": kms["k_zfft"] / zl,
                  "alg_bytes_per_launch": per_launch,
                  "pmc_bytes_per_launch": (traffic_file or {}).get("bytes_per_launch", {}).get("k_zfft")}
            dom_kernel, dom_launches = "k_zfft (pacing member of z_stage)", zl
        isolated = None
        if iso is not None:
            ims = iso["kernel_ms"]
            isolated = {k: {"ms_per_step": ims[k]} for k in ims}
            for k in ("k_yfft", "k_xfft"):
                if ims[k] > 0:
                    isolated[k]["alg_GBps"] = alg[k] * particles / (ims[k] * 1e-3) / 1e9
        out = {
            "metric": "particles/sec (grid->displacements)", "value": value, "unit": "particles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (seeded pcg64 Gaussian modes, wmap1new P(k)%s)" % (
                ", synthetic ppd_e=128 PLT eigenmodes" if plt else ""),
            "config": {"workload": "PPD=%d %s ICFormat=%s seed=12346 BoxSize=720" % (
                N, "ZD_qPLT=1 ZD_qPLT_rescale=1" if plt else "ZA (ZD_qPLT=0)", fmt),
                "stream_factor": R, "narray_reference": narray,
                "store": ("fields E,Z of the half-space rows, zero columns not stored" if fields else
                          "%d arrays" % plan.narray), "passes": plan.passes,
                "block_store_GB": plan.exchange_bytes / 1e9,
                "parallelism": ("%d group(s) of %d GPU(s): residue passes round-robin over the groups%s" % (
                    groups, gsz, (", ky/z slabs + one RCCL exchange per pass inside a group" + (", passes pipelined over two send stores" if pipelined else ""))
                    if gsz > 1 else ", no exchange"))},
            "hbm_GBps_path": (64.0 * narray + recsize) * value / 1e9,
            "roofline_path_frac": (64.0 * narray + recsize) * value / 1e9 / (HBM_PEAK_GBS * world),
            # `bound`: the y / x stages and the packed-PLT Z stage wait for HBM; the ZA Z stage (generator || z FFT) is bound by
            # vector instructions (SQ counters of the committed profile: three generator waves per SIMD x 33.5 % issue) — its
            # `achieved` stays the SURVEY 8d byte figure of merit, `valu_issue` carries the counters when the profile has them
            "roofline": {"bound": "valu" if (dom == "z_stage" and not plt) else "hbm", "kernel": dom_kernel, "achieved": du["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": du["alg_GBps"] / HBM_PEAK_GBS, "traffic": du.get("pmc_bytes_per_launch"),
                         "avg_launch_ms": du["avg_launch_ms"], "launches": dom_launches,
                         "alg_bytes_per_launch": du["alg_bytes_per_launch"],
                         "alg_bytes_per_particle": alg[dom],
                         "note": "achieved = SURVEY 8d algorithmic bytes of this unit per launch / hipEvent launch time "
                                 "(launch stream, timed region): a figure of merit against the reference's traffic, not "
                                 "bytes moved — this implementation moves fewer (design_bytes_per_step; PMC in traffic)"
                                 + ("; the dominant unit is the Z stage = generator || z FFT on two streams, quoted on the z FFT "
                                    "that paces it; the stage is bound by the two kernels' vector work (pcg64, Box-Muller, P(k) per "
                                    "mode), not by HBM: kernels.k_yfft / k_xfft are the HBM-bound units"
                                    if dom == "z_stage" and not plt else "")},
            "kernels": per_kernel,
            "kernels_isolated": isolated,
            "source_sha": source_sha(),
        }
        if traffic_stale:
            out["roofline"]["traffic_stale"] = traffic_stale
        if out["roofline"]["bound"] == "valu":
            out["roofline"]["valu_issue"] = (traffic_file or {}).get("sq")  # per kernel: VALU-issue cycles / wave cycles, waves per SIMD
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["exchange"] = {"transport": "RCCL grouped ncclSend/ncclRecv per plane group (zd_plan_run_pass)",
                               "GB_per_step_all_ranks": sum(r["GB_sent_per_step"] for r in per_rank),
                               "max_exchange_wait_ms": max(r["exchange_wait_ms"] for r in per_rank)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, plt, fmt, eig)
            except Exception as e:  # the baseline is a reported number, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "particles/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
    # N > 1 (or --dist): the OTHER way to share the job is timed in the same run, so that one SCALE record holds both — BASELINE C4
    # names the slab all-to-all ("one group of all GPUs"), the library's default for PPD=4096 is one GPU per pass group with no
    # exchange at all; `value` is the default's, `modes` has both with per-rank spans and bytes
    modes = None
    if multi:
        # a split is named by what happens between its GPUs: ranks of a group exchange (all_to_all) or every GPU is its own group
        first_name = "all_to_all" if gsz > 1 else "pass_groups"
        other_name = "pass_groups" if first_name == "all_to_all" else "all_to_all"
        modes = {"default": first_name, first_name: mode_summary(main_mode),
                 # what the default was chosen on: the probed rate of one link (all links of a GPU busy) and the library's estimate
                 # of a step for both splits at that rate (empty: only one split exists for this job, or nothing was measured)
                 "link_probe": link, "estimate_at_probed_rate": choice_est}
        close_mode(main_mode)
        plan = comm = pipe = None
        other = None
        # ... also when it never returns (a collective that hangs): every rank arms the same deadline, rank 0 prints the line it has
        import threading
        deadline = 60.0 + 6.0 * (time.perf_counter() - t_start)

        def give_up():
            if rank == 0:
                modes[other_name] = {"error": "no result within %.0f s of starting it; the process was ended" % deadline}
                out["modes"] = modes
                out["aborted"] = True
                print(json.dumps(out), flush=True)
            os._exit(3)  # the line is kept, the status says that this run did not end by itself

        watchdog = threading.Timer(deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
        # ... and when the process is killed inside it (a fault in a collective that has never run on this hardware, or the launcher's
        # SIGTERM after another rank died): rank 0 leaves the line it has from a C-level signal handler — a Python-level one would not
        # run while the main thread sits in a C call
        last_words = None
        if rank == 0:
            def line_for(sig):
                m2 = dict(modes)
                m2[other_name] = {"error": "the process received signal %d while measuring this split" % sig}
                return json.dumps(dict(out, modes=m2, aborted=True))

            last_words = arm_last_words(line_for)
        try:  # (the line with `value` must be printed whatever happens to the second measurement)
            if os.environ.get("ZD_BENCH_SELFTEST_SIGNAL"):  # tests/test_gpu_multi_rehearsal.py: what the launcher's SIGTERM would do here
                os.kill(os.getpid(), int(os.environ["ZD_BENCH_SELFTEST_SIGNAL"]))
                time.sleep(30)
            other = run_split(world if first_name == "all_to_all" else 1)  # one GPU per group <-> one group of all GPUs
            if other is not None and (world == 1 or (other["gsz"] > 1) != (gsz > 1)):
                modes[other_name] = mode_summary(other)
            else:
                modes[other_name] = {"unavailable": "the passes of this workload do not deal out that way on %d GPU(s)" % world}
        except Exception as e:
            modes[other_name] = {"error": repr(e)}
        watchdog.cancel()
        if last_words is not None:
            last_words[1]()
        if other is not None:
            try:
                close_mode(other)
            except Exception:
                pass
    if rank == 0:
        if modes is not None:
            out["modes"] = modes
        print(json.dumps(out))
    if plan is not None:
        plan.close()
    if comm is not None:
        comm.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
