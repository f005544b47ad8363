"""TEST-ONLY stand-in for the staged engine interface (include/zeldovich_hip.h, "staged API"): it
implements the documented block-store layout (chunks per destination rank, planes outermost inside a chunk) and the
CYCLIC rank ownership of the half-space rows (ky = rank + world*i) with numpy + the oracle's mode cube, so the
one-process-per-rank driver (zeldovich_plt_amd/parallel.py) — exchange in plane groups into a two-slot ring, XY stages
per group — can be exercised on CPU with gloo.  It is never imported by the product."""
import numpy as np
import torch


class NumpyEngine:
    def __init__(self, cube, ppd, R, rank, world):
        """cube: oracle mode cube [a][ky][kz][kx] (full Hermitian cube, Nyquist row zero)"""
        self.N, self.R, self.rank, self.world = ppd, R, rank, world
        self.na = cube.shape[0]
        N = ppd
        self.L = N // R
        self.Hq = N // 2 // world
        self.Zq = self.L // world
        self.local_planes = self.Zq
        self.passes = R
        self.plane_step = 1
        self.record_size = 16 * self.na  # "records" of the stand-in = the raw complex planes
        # block store as the library lays it out (zd_device.h StoreLayout with the library's defaults): inside a chunk
        # [plane][array][row slot][x] — planes outermost, so a plane group of every chunk is contiguous
        self.a_stride = 2 * self.Hq * N
        self.z_stride = self.a_stride * self.na
        self.exchange_bytes = self.z_stride * self.Zq * world * 16
        # z-transformed columns for the rows this rank generates
        self.zt = np.fft.ifft(cube, axis=2) * N  # [a][ky][z][kx]

    def plane_z(self, residue, local_plane):
        return residue + self.R * (self.rank * self.Zq + local_plane)

    def _loc(self, ky):  # (source rank, row slot) — zd_device.h row_slot: CYCLIC ownership, ky = rank + world*i
        N, G = self.N, self.world
        if ky < N // 2:
            kyh, tw = ky, 0
        elif ky == N // 2:
            kyh, tw = 0, 1
        else:
            kyh, tw = N - ky, 1
        return kyh % G, kyh // G + tw * self.Hq

    def _off(self, chunk, zl, a, slot, chunk_planes):
        return ((chunk * chunk_planes + zl) * self.na + a) * self.a_stride + slot * self.N

    def stage_z(self, residue, send):
        buf = send.numpy().view(np.complex128).reshape(-1)
        N, Hq, Zq, G = self.N, self.Hq, self.Zq, self.world
        for i in range(Hq):
            kyh = self.rank + G * i
            rows = [(kyh, i)]
            if kyh != 0:
                rows.append((N - kyh, Hq + i))
            for ky, loc in rows:
                for z2 in range(self.L):
                    dst, zl = divmod(z2, Zq)
                    for a in range(self.na):
                        o = self._off(dst, zl, a, loc, Zq)
                        buf[o:o + N] = self.zt[a, ky, residue + self.R * z2, :]

    def stage_y_group(self, recv, chunk_planes, nplanes):
        buf = recv.numpy().view(np.complex128).reshape(-1)
        N = self.N
        for zl in range(nplanes):
            for a in range(self.na):
                offs = []
                plane = np.zeros((N, N), dtype=np.complex128)
                for ky in range(N):
                    src, loc = self._loc(ky)
                    o = self._off(src, zl, a, loc, chunk_planes)
                    offs.append(o)
                    if ky != N // 2:
                        plane[ky] = buf[o:o + N]
                plane = np.fft.ifft(plane, axis=0) * N
                for y in range(N):
                    buf[offs[y]:offs[y] + N] = plane[y]

    def stage_x_group(self, residue, recv, chunk_planes, plane0, gplane0, nplanes, out):
        buf = recv.numpy().view(np.complex128).reshape(-1)
        o_out = out.numpy().view(np.complex128)
        N = self.N
        for i in range(nplanes):
            zl = plane0 + i
            for a in range(self.na):
                for y in range(N):
                    src, loc = self._loc(y)
                    o = self._off(src, zl, a, loc, chunk_planes)
                    row = np.fft.ifft(buf[o:o + N]) * N
                    d = ((i * N + y) * self.na + a) * N  # out layout: [plane][y][a][x]
                    o_out[d:d + N] = row
