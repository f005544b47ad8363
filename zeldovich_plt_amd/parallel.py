"""One-process-per-GPU driver of the grid->displacements path.

The reference is single-process; its y<->z "transpose" is StoreBlock/LoadBlock on one big array
(src/block_array.cpp:387-414,466-504).  Here rank g of G owns the half-space rows ky = g, g + G, g + 2G, ...
(cyclic: the rows near ky = 0 carry most of the non-zero modes) during the Z stage and the planes
[g*Zq,(g+1)*Zq) of each residue pass during the XY stage; the block exchange between the two is ONE all-to-all per pass (RCCL over xGMI through
torch.distributed; gloo in the CPU tests).  No other collective is on the data path.

`engine` is anything with the staged interface of include/zeldovich_hip.h:
    engine.R, engine.passes, engine.plane_step, engine.local_planes, engine.exchange_bytes, engine.record_size
    engine.stage_z(residue, send), engine.stage_y(recv), engine.stage_x(residue, recv, p0, n, out)
    engine.plane_z(residue, local_plane)
On a GPU it is `HipEngine` (zeldovich_plt_amd.api.Plan on torch device buffers).
"""
import torch


class HipEngine:
    """The product engine: libzeldovich_hip.so plan on torch-owned HBM buffers (no CPU fallback)."""

    def __init__(self, plan, ppd):
        self.plan = plan
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

    def stage_z(self, residue, send):
        self.plan.stage_z(residue, send.data_ptr(), self._stream())

    def stage_y(self, recv):
        self.plan.stage_y(recv.data_ptr(), self._stream())

    def stage_x(self, residue, recv, plane0, nplanes, out):
        self.plan.stage_x(residue, recv.data_ptr(), plane0, nplanes, out.data_ptr(), None, self._stream())


class SlabPipeline:
    """Runs residue passes: Z stage -> all-to-all -> y FFT -> x FFT + epilogue in plane chunks."""

    def __init__(self, engine, ppd, world=1, dist=None, device="cpu", chunk_bytes=8 << 30):
        # chunk_bytes: size of the record ring = planes finished per x-stage launch.  At PPD=4096 one plane of RVZel
        # records is 537 MB; launches of a single store plane (4096 workgroups, 16 per CU) lose 13 % to launch tails
        # (k_xfft 0.89 -> 0.77 s per step with 8 GB).
        self.e = engine
        self.ppd = ppd
        self.world = world
        self.dist = dist
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs torch.distributed")
        # float64 elements: a dtype every backend (RCCL, gloo) moves natively, 8x fewer elements than bytes
        assert engine.exchange_bytes % (8 * world) == 0
        nel = engine.exchange_bytes // 8
        self.send = torch.empty(nel, dtype=torch.float64, device=device)
        self.recv = torch.empty(nel, dtype=torch.float64, device=device) if world > 1 else self.send
        plane_b = ppd * ppd * max(engine.record_size, 1)
        step = getattr(engine, "plane_step", 1)  # a store plane may deliver two z planes (packed ZA store)
        self.chunk = int(max(step, min(engine.local_planes, chunk_bytes // plane_b) // step * step))
        self.ring = torch.empty(self.chunk * plane_b, dtype=torch.uint8, device=device)

    # largest single message of the exchange, in float64 elements: at PPD=4096 on 2 ranks a peer's chunk is
    # 52 GB = 6.5e9 elements — past what a 32-bit element count anywhere inside a collective library could hold.
    # Below the limit (e.g. 8 ranks: 1.6e9) the exchange is ONE all_to_all_single.
    MAX_MSG_ELEMS = (1 << 31) - 1

    def exchange(self):
        if self.world <= 1:
            return
        # chunk d of `send` goes to rank d and arrives as chunk <my rank> there: exactly the
        # y-slab-owner -> z-slab-owner block move of StoreBlock/LoadBlock
        per = self.send.numel() // self.world
        if per <= self.MAX_MSG_ELEMS:
            self.dist.all_to_all_single(self.recv, self.send)
            return
        # same move in k rounds of equal contiguous pieces (piece i of every peer chunk per round)
        k = -(-per // self.MAX_MSG_ELEMS)
        while per % k:
            k += 1
        part = per // k
        sv, rv = self.send.view(self.world, k, part), self.recv.view(self.world, k, part)
        me = self.dist.get_rank()
        for i in range(k):
            ops = []
            for d in range(self.world):
                if d == me:
                    rv[d, i].copy_(sv[d, i])
                else:
                    ops.append(self.dist.P2POp(self.dist.isend, sv[d, i], d))
                    ops.append(self.dist.P2POp(self.dist.irecv, rv[d, i], d))
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()

    def run_pass(self, residue, consume=None):
        """consume(z_list, ring_tensor) is called once per plane chunk with the global z of each plane"""
        e = self.e
        e.stage_z(residue, self.send)
        self.exchange()
        e.stage_y(self.recv)
        for p0 in range(0, e.local_planes, self.chunk):
            n = min(self.chunk, e.local_planes - p0)
            e.stage_x(residue, self.recv, p0, n, self.ring)
            if consume is not None:
                consume([e.plane_z(residue, p0 + i) for i in range(n)], self.ring)

    def run(self, consume=None):
        for r in range(getattr(self.e, "passes", self.e.R)):
            self.run_pass(r, consume)
