"""One-process-per-GPU driver of the grid->displacements path.

The reference is single-process; its y<->z "transpose" is StoreBlock/LoadBlock on one big array
(src/block_array.cpp:387-414,466-504).  Here rank g of G owns the half-space rows ky = g, g + G, g + 2G, ...
(cyclic: the rows near ky = 0 carry most of the non-zero modes) during the Z stage and the planes
[g*Zq,(g+1)*Zq) of each residue pass during the XY stage.  The send store is laid out
[destination rank][plane][array|field][row slot][x]: the planes [p0, p1) of every chunk are contiguous, so the
exchange runs in PLANE GROUPS into a two-slot ring — group j+1 travels while the y and x stages of group j run —
and no second full-size store exists.  No other collective is on the data path.

The residue passes of the z streaming are independent partitions of the output, so the ranks may also work as several
GROUPS (`split_ranks`, zd_choose_pass_groups): group j runs the passes j, j + groups, ... and nothing travels between groups;
with one rank per group there is no exchange at all.

On a GPU the whole pass (Z stage, exchange over RCCL/xGMI, XY stages) runs inside the library
(`zd_plan_run_pass`, csrc/zd_multi.cpp); `SlabPipeline` then only owns the buffers.  The same pipeline written against
an abstract `engine` (anything with the staged interface of include/zeldovich_hip.h) and torch.distributed
point-to-point operations is what the CPU tests drive with gloo and a numpy stand-in engine:
    engine.R, .passes, .plane_step, .local_planes, .exchange_bytes, .record_size, .world
    engine.stage_z(residue, send), engine.stage_y_group(buf, chunk_planes, nplanes),
    engine.stage_x_group(residue, buf, chunk_planes, plane0, gplane0, nplanes, out), engine.plane_z(residue, local_plane)
"""
import torch


def split_ranks(rank, world, groups):
    """(group, rank inside the group, ranks per group) of global rank `rank` when `world` ranks work as `groups` groups of
    consecutive ranks (the split zd_generate_multi uses for its host threads)"""
    assert world % groups == 0
    gsz = world // groups
    return rank // gsz, rank % gsz, gsz


class HipEngine:
    """The product engine: libzeldovich_hip.so plan on torch-owned HBM buffers (no CPU fallback)."""

    native = True

    def __init__(self, plan, ppd, comm=None):
        self.plan = plan
        self.comm = comm
        self.ppd = ppd
        self.R = plan.R
        self.passes = plan.passes
        self.plane_step = plan.plane_step
        self.local_planes = plan.local_planes
        self.exchange_bytes = plan.exchange_bytes
        self.record_size = plan.record_size

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def plane_z(self, residue, local_plane):
        return self.plan.plane_z(residue, local_plane)

    def run_pass(self, residue, store, ring, rec_planes, consume=None):
        """Z stage -> exchange in plane groups (overlapped) -> y / x stages, all inside the library"""
        self.plan.run_pass(residue, store.data_ptr(), ring.data_ptr(), rec_planes, comm=self.comm, consume=consume,
                           stream=self._stream())


def _hip_run_passes(engine, first, step, store, store2, ring, rec_planes, consume=None):
    engine.plan.run_passes(first, step, store.data_ptr(), None if store2 is None else store2.data_ptr(), ring.data_ptr(), rec_planes,
                           comm=engine.comm, consume=consume, stream=engine._stream())


class SlabPipeline:
    """Runs residue passes: Z stage -> exchange of plane groups -> y FFT -> x FFT + epilogue per group."""

    def __init__(self, engine, ppd, world=1, dist=None, device="cpu", chunk_bytes=8 << 30, group_bytes=4 << 30, rank_base=0,
                 process_group=None):
        # world = ranks of THIS group (they exchange); rank_base = global rank of the group's rank 0, process_group = its
        # torch.distributed group (stand-in path only; None: the default group)
        # chunk_bytes: size of the record ring = planes finished per x-stage launch.  At PPD=4096 one plane of RVZel
        # records is 537 MB; launches of a single store plane (4096 workgroups, 16 per CU) lose 13 % to launch tails.
        self.e = engine
        self.ppd = ppd
        self.world = world
        self.dist = dist
        self.rank_base = rank_base
        self.pg = process_group
        self.native = bool(getattr(engine, "native", False))
        if world > 1 and dist is None and not self.native:
            raise ValueError("world > 1 needs torch.distributed")
        assert engine.exchange_bytes % (8 * world) == 0
        nel = engine.exchange_bytes // 8
        # float64 elements: a dtype every backend moves natively
        self.send = torch.empty(nel, dtype=torch.float64, device=device)
        self.send2 = None  # second send store: pipelined passes of ranks that exchange (alloc_second_store)
        plane_b = ppd * ppd * max(engine.record_size, 1)
        step = getattr(engine, "plane_step", 1)  # a store plane may deliver two z planes (packed ZA stores)
        self.chunk = int(max(step, min(engine.local_planes, chunk_bytes // plane_b) // step * step))
        self.ring = torch.empty(self.chunk * plane_b, dtype=torch.uint8, device=device)
        # stand-in path only: receive ring of two slots, each [source rank][group planes] of a chunk
        self.Zq = engine.local_planes // step
        self.chunk_plane_el = nel // world // self.Zq  # float64 elements of one store plane inside one chunk
        if world > 1 and not self.native:
            gp = max(1, min(self.Zq, group_bytes // (self.chunk_plane_el * 8 * world)))
            self.group_planes = int(gp)
            self.recv = torch.empty(2 * world * gp * self.chunk_plane_el, dtype=torch.float64, device=device)

    # ---- exchange of one plane group (stand-in path): chunk <me> of every peer, planes [p0, p0+np) -> ring slot ----
    def _exchange_group(self, j):
        w, gp, cpe = self.world, self.group_planes, self.chunk_plane_el
        p0 = j * gp
        npl = min(gp, self.Zq - p0)
        slot = self.recv[(j & 1) * w * gp * cpe:((j & 1) + 1) * w * gp * cpe].view(w, gp * cpe)
        sv = self.send.view(w, self.Zq * cpe)
        me = self.dist.get_rank() - self.rank_base
        ops = []
        for d in range(w):
            src = sv[d, p0 * cpe:(p0 + npl) * cpe]          # what rank d will finish
            dst = slot[d, :npl * cpe]                        # what rank d generated for me
            if d == me:
                dst.copy_(src)
            else:
                ops.append(self.dist.P2POp(self.dist.isend, src, self.rank_base + d))
                ops.append(self.dist.P2POp(self.dist.irecv, dst, self.rank_base + d))
        return (self.dist.batch_isend_irecv(ops) if ops else []), slot, npl

    def run_pass(self, residue, consume=None):
        """consume(z_list, ring_tensor) is called once per plane chunk with the global z of each plane"""
        e = self.e
        step = getattr(e, "plane_step", 1)
        if self.native:
            cb = None
            if consume is not None:
                def cb(first, n, recp, st):
                    consume([e.plane_z(residue, first + i) for i in range(n)], self.ring)
                    return 0
            e.run_pass(residue, self.send, self.ring, self.chunk, cb)
            return
        e.stage_z(residue, self.send)
        if self.world <= 1:
            e.stage_y_group(self.send, self.Zq, self.Zq)
            for p0 in range(0, e.local_planes, self.chunk):
                n = min(self.chunk, e.local_planes - p0)
                e.stage_x_group(residue, self.send, self.Zq, p0, p0, n, self.ring)
                if consume is not None:
                    consume([e.plane_z(residue, p0 + i) for i in range(n)], self.ring)
            return
        gp = self.group_planes
        ngroups = -(-self.Zq // gp)
        pending = self._exchange_group(0)
        for j in range(ngroups):
            reqs, slot, npl = pending
            if j + 1 < ngroups:
                pending = self._exchange_group(j + 1)  # the next group travels while this one is transformed
            for r in reqs:
                r.wait()
            buf = slot.view(-1)
            e.stage_y_group(buf, gp, npl)
            for q0 in range(0, npl * step, self.chunk):
                n = min(self.chunk, npl * step - q0)
                g0 = j * gp * step + q0
                e.stage_x_group(residue, buf, gp, q0, g0, n, self.ring)
                if consume is not None:
                    consume([e.plane_z(residue, g0 + i) for i in range(n)], self.ring)
        # every peer has taken its planes before the next Z stage overwrites the send store
        self.dist.barrier(group=self.pg)

    def alloc_second_store(self):
        """native path, several ranks per group: a second send store lets the library pipeline the passes (Z stage of pass
        p + 1 beside the exchange of pass p, zd_plan_run_passes)"""
        if self.native and self.send2 is None:
            self.send2 = torch.empty_like(self.send)

    def run(self, consume=None, pass_first=0, pass_step=1):
        """all passes, or (several groups of ranks) the passes pass_first, pass_first + pass_step, ... of this group"""
        if self.native and consume is None:
            _hip_run_passes(self.e, pass_first, pass_step, self.send, self.send2, self.ring, self.chunk)
            return
        for r in range(pass_first, getattr(self.e, "passes", self.e.R), pass_step):
            self.run_pass(r, consume)
