"""TEST-ONLY stand-in for the staged engine interface (include/zeldovich_hip.h, "staged API"): it
implements the documented block-store layout and rank ownership with numpy + the oracle's mode cube,
so the one-process-per-rank driver (zeldovich_plt_amd/parallel.py) and its all-to-all can be
exercised on CPU with gloo.  It is never imported by the product."""
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
        self.record_size = 16 * self.na  # "records" of the stand-in = the raw complex planes
        # tiled block store (zd_device.h StoreLayout): blocks of Bz planes x Bk row slots x N
        target = max(1, (2 << 20) // (N * 16))
        lt = target.bit_length() - 1
        lBk = (lt + 1) // 2
        lBz = lt - lBk
        while (1 << lBk) > self.Hq:
            lBk -= 1
        while (1 << lBz) > self.Zq:
            lBz -= 1
        self.lBk, self.lBz = lBk, lBz
        lBk = min(20, self.Hq.bit_length() - 1)  # library default: all row slots of a half in one block,
        lBz = 0                                   # one plane per block (measured best on MI355X)
        self.lBk, self.lBz = lBk, lBz
        self.a_stride = (N << lBk) << lBz
        self.zb_stride = self.a_stride * self.na
        self.kb_stride = self.zb_stride * (self.Zq >> lBz)
        self.chunk_stride = self.kb_stride * ((2 * self.Hq) >> lBk)
        self.exchange_bytes = self.chunk_stride * world * 16
        # z-transformed columns for the rows this rank generates
        self.zt = np.fft.ifft(cube, axis=2) * N  # [a][ky][z][kx]

    def plane_z(self, residue, local_plane):
        return residue + self.R * (self.rank * self.Zq + local_plane)

    def _loc(self, ky):  # (source rank, row slot) — zd_device.h row_offset
        N, Hq = self.N, self.Hq
        if ky < N // 2:
            kyh, tw = ky, 0
        elif ky == N // 2:
            kyh, tw = 0, 1
        else:
            kyh, tw = N - ky, 1
        src = kyh // Hq
        return src, kyh - src * Hq + tw * Hq

    def _off(self, chunk, zl, a, slot):
        Bk, Bz = 1 << self.lBk, 1 << self.lBz
        return (chunk * self.chunk_stride + (slot >> self.lBk) * self.kb_stride + (zl >> self.lBz) * self.zb_stride
                + a * self.a_stride + (((zl & (Bz - 1)) << self.lBk) + (slot & (Bk - 1))) * self.N)

    def stage_z(self, residue, send):
        buf = send.numpy().view(np.complex128).reshape(-1)
        N, Hq, Zq = self.N, self.Hq, self.Zq
        for kyh in range(self.rank * Hq, (self.rank + 1) * Hq):
            rows = [(kyh, kyh - self.rank * Hq)]
            if kyh != 0:
                rows.append((N - kyh, Hq + kyh - self.rank * Hq))
            for ky, loc in rows:
                for z2 in range(self.L):
                    dst, zl = divmod(z2, Zq)
                    for a in range(self.na):
                        o = self._off(dst, zl, a, loc)
                        buf[o:o + N] = self.zt[a, ky, residue + self.R * z2, :]

    def stage_y(self, recv):
        buf = recv.numpy().view(np.complex128).reshape(-1)
        N = self.N
        for zl in range(self.Zq):
            for a in range(self.na):
                offs = []
                plane = np.zeros((N, N), dtype=np.complex128)
                for ky in range(N):
                    src, loc = self._loc(ky)
                    o = self._off(src, zl, a, loc)
                    offs.append(o)
                    if ky != N // 2:
                        plane[ky] = buf[o:o + N]
                plane = np.fft.ifft(plane, axis=0) * N
                for y in range(N):
                    buf[offs[y]:offs[y] + N] = plane[y]

    def stage_x(self, residue, recv, plane0, nplanes, out):
        buf = recv.numpy().view(np.complex128).reshape(-1)
        o_out = out.numpy().view(np.complex128)
        N = self.N
        for i in range(nplanes):
            zl = plane0 + i
            for a in range(self.na):
                for y in range(N):
                    src, loc = self._loc(y)
                    o = self._off(src, zl, a, loc)
                    row = np.fft.ifft(buf[o:o + N]) * N
                    d = ((i * N + y) * self.na + a) * N  # out layout: [plane][y][a][x]
                    o_out[d:d + N] = row
