// zd_kernels_fz.hip — the Z stage of the packed PLT store (PACK_PLT3) as ONE kernel: generator + z FFT + Hermitian stores, the folded
// inputs never leave the CU (round 5).
//
// Replaces, for the half-space rows ky >= 1: the (z, x) loop of LoadPlane with cgauss<2> and get_eigenmode
// (src/zeldovich.cpp:333-438, src/power_spectrum.cpp:338-359, src/zeldovich.cpp:154-276), the packing and the twins (:447-466),
// InverseFFT_Yonly (:508-511) and StoreBlock (src/block_array.cpp:387-414) — what k_genf -> Y -> k_zfft do in two kernels with an
// HBM round trip of the folded inputs between them (PPD=2048 PLT, BASELINE C3: 0.80 of the stage's 1.22 TB).
//
// Orientation.  k_genf puts a wave's lanes along x and lets a thread walk z; its output is written [k2][x] and read back by a z FFT
// whose lines run along k2.  Here the lanes run ALONG kz and a workgroup walks a row along kx:
//   * a workgroup of L/2 threads owns one half-space row ky and a range of columns; thread t owns, for every column, the two line
//     elements k2 = t and k2 = L - t (thread 0: k2 = 0 and L/2) with all their fold terms — the modes kz = +-t and +-(L - t) at
//     R = 2, kz = +-t at R = 1: exactly R mirror pairs (kz, -kz), which share |k|^2, the zero rule, P(k), 1/k^2, the eigenmode,
//     f and the rescale factor (the kz mirror of genf_tile_kz, without a second walker);
//   * a mode's two draws sit at counter 2 ((ky 65536 + (kz mod 65536)) 65536 + (kx mod 65536)): for fixed (ky, kz) the columns
//     kx, kx + 1, ... are CONSECUTIVE draws, so a thread's 2R generators just step — no jump maps; the start states of a work
//     item come from the 2^i table once per item (<= 128 columns);
//   * everything that depends on (ky, kz) only — kz^2, the fold twiddles, the eigenmode table's (y, z) corner blend — is hoisted out
//     of the column loop; the (y, z)-blended table entries of the two x cells are re-read when kx leaves the cell (every N / ppd_e
//     columns), so the hot loop has no global loads at all;
//   * the 6 job lines of the column (L points each, 96 KB at L = 1024) are staged in LDS; six of the eight waves then own one
//     line each (lane t holds k2 = t + 64 e: the register layout of zd_fft.h), transform it with wave-private exchanges — no
//     workgroup barrier inside the transform — and store it: self jobs at (row ky, column x), twin jobs conjugated at
//     (row N - ky, column N - x).
// Stores.  A lane holds planes t + 64 e of ONE column: with the plain [plane][array][row][x] store an instruction would write 64
// pieces of 16 bytes (measured 1.1 TB/s; two columns side by side, all a CU can hold: 32-byte runs, 1.7 TB/s —
// scripts/microbench/fused_plt_store.hip).  The store is therefore laid out with 4 planes interleaved along x
// (StoreLayout::lq = 2): lanes 4i .. 4i + 3 write one aligned 64-byte run (4.8 TB/s), and the y / x stages read whole lines.
// Work items (row, first column, columns) are built on the host from the zero rule (dead columns are never visited) and pulled
// from an atomic counter, longest first.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>

#include "zd_device.h"
#include "zd_launch.h"
#include "zd_genmath.h"

using namespace zd;
using namespace zdgen;
using zdfft::cplx;
using zdpcg::u128;

__constant__ zdpcg::BitTable c_bits_fz;

extern "C" int zdk_upload_bit_table_fz(const zdpcg::BitTable *host) {
    return (int) hipMemcpyToSymbol(HIP_SYMBOL(c_bits_fz), host, sizeof(zdpcg::BitTable));
}

namespace {

__device__ __forceinline__ u128 advance_bits_fz(u128 s, uint64_t delta) {
    for (int i = 0; i < zdpcg::NBITS; i++) {
        if ((delta >> i) == 0) break;
        if ((delta >> i) & 1ULL) s = zdpcg::apply(c_bits_fz.m[i], s);
    }
    return s;
}

// LDS traffic of ONE wave between two phases of its private exchange: the LDS queue of a wave is in order, so a read issued after
// a write sees it; what has to be stopped is the compiler moving accesses across
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// zdfft::fft_line for a line owned by ONE wave (PL::T == 64): the same passes, the exchanges without workgroup barriers
template <class PL, class XL, int P = 0>
__device__ __forceinline__ void fft_line_wave(double (&re)[PL::E], double (&im)[PL::E], int t, double *lds, const cplx *__restrict__ tw) {
    zdfft::pass_compute<PL, P>(re, im, t, tw);
    if constexpr (P + 1 < PL::NPASS) {
        zdfft::xchg_write<PL, P, XL>(re, t, 0, lds);
        wave_lds_sync();
        zdfft::xchg_read<PL, XL>(re, t, 0, lds);
        wave_lds_sync();
        zdfft::xchg_write<PL, P, XL>(im, t, 0, lds);
        wave_lds_sync();
        zdfft::xchg_read<PL, XL>(im, t, 0, lds);
        wave_lds_sync();
        fft_line_wave<PL, XL, P + 1>(re, im, t, lds, tw);
    }
}

// (y, z)-blended table entries of x cell `cx` for a thread's |kz|: sum over the 4 (y, z) corners of w_y w_z E[cx][cy][cz][0..3]
// (zero-weight corners are not read, like get_eigenmode_dev)
__device__ __forceinline__ void eig_yz_blend(const GenConst &g, int cx, const EigAxis &ay, const EigAxis &az, double (&out)[4]) {
    const int ep = (int) g.eig_ppd, halfppd = ep / 2 + 1;
    const double2 *E = reinterpret_cast<const double2 *>(g.eig);
    out[0] = out[1] = out[2] = out[3] = 0.0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const double w = ((c & 2) ? ay.f : 1 - ay.f) * ((c & 1) ? az.f : 1 - az.f);
        if (w != 0) {
            const int i = ((cx * ep + ((c & 2) ? ay.h : ay.l)) * halfppd + ((c & 1) ? az.h : az.l)) * 2;
            const double2 q0 = E[i], q1 = E[i + 1];
            out[0] += w * q0.x;
            out[1] += w * q0.y;
            out[2] += w * q1.x;
            out[3] += w * q1.y;
        }
    }
}

}  // namespace

// grid: persistent, one workgroup per CU (the staged lines + the generator's table image are 118 KB of LDS)   block: L / 2
template <int L, int R, bool PLAW>
__global__ __launch_bounds__(L / 2) void k_genz_plt(GenConst g, StoreLayout S, int ky0, int residue, unsigned nitems,
                                                   const FzItem *__restrict__ items, const cplx *__restrict__ twN,
                                                   const cplx *__restrict__ twL, cplx *__restrict__ out,
                                                   unsigned *__restrict__ ctr) {
    static_assert(L == 1024 && (R == 1 || R == 2), "one wave per 1024-point line; one or two fold terms");
    constexpr int NT = L / 2, E = 16, NJOB = 6;
    using PL = zdfft::Plan<L, E>;
    using XL = zdfft::LineInner<L, 1>;
    static_assert(PL::T == 64, "a line is one wave");
    extern __shared__ __attribute__((aligned(16))) double T[];  // GenfTab image | work-item slot | staged lines [job][k2]
    for (int i = threadIdx.x; i < g.genf_n / 2; i += NT) reinterpret_cast<double2 *>(T)[i] = reinterpret_cast<const double2 *>(g.genf_tab)[i];
    unsigned *slot = reinterpret_cast<unsigned *>(T + g.genf_n);
    cplx *stage    = reinterpret_cast<cplx *>(T + ((g.genf_n + 3) & ~1));
    const int N = g.N, half = g.half;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;

    // ---- what depends on the thread only ----
    // pair p of the thread: |kz| = a[p]; its "+" mode is fold term 0 of line element a[p], its "-" mode fold term R - 1 of element
    // L - a[p].  Line elements owned: kA = t, kB = L - t (thread 0: 0 and L/2)
    int a[R];
    a[0] = t;
    if constexpr (R == 2) a[1] = t == 0 ? L / 2 : L - t;
    const int kA = t, kB = t == 0 ? L / 2 : L - t;
    EigAxis az[R];
#pragma unroll
    for (int p = 0; p < R; p++) az[p] = eig_axis(g, a[p]);  // the table's +kz half space
    // final factor W_N^{k2 residue} of the two line elements; fold factor W_R^{(R-1) residue} of the "-" modes
    double wAr = 1.0, wAi = 0.0, wBr = 1.0, wBi = 0.0;
    if (R > 1) {
        const cplx wa = twN[(kA * residue) & (N - 1)], wb = twN[(kB * residue) & (N - 1)];
        wAr = wa.x, wAi = wa.y, wBr = wb.x, wBi = wb.y;
    }
    const double fold_m = (R == 2 && (residue & 1)) ? -1.0 : 1.0;
    // thread 0: pair 0 is the single mode kz = 0; at R = 2 its pair 1 (kz = +-L/2) puts BOTH modes on element kB
    const double to_a = t != 0 ? 1.0 : 0.0, to_b = 1.0 - to_a;
    const bool has_m0 = t != 0;

    double vsum = 0.0;
    for (;;) {
        __syncthreads();  // table image complete / previous item's slot consumed / previous column's exchanges done
        if (threadIdx.x == 0) *slot = atomicAdd(ctr, 1u);
        __syncthreads();
        const unsigned item = *slot;
        if (item >= nitems) break;
        const FzItem it = items[item];
        const int kyl = it.kyl, ky = ky0 + kyl * S.ky_stride;
        const EigAxis ay = eig_axis(g, ky);
        const bool dead_y = ky == g.kmax;
        // generators: state ONE draw ahead of the first column's counter (zd_kernels.hip genf_tile)
        u128 sp[R], sm[R];
        {
            const int kx0 = it.x0 > half ? it.x0 - N : it.x0;
#pragma unroll
            for (int p = 0; p < R; p++) {
                const uint64_t cx = (uint64_t) (kx0 & 65535);
                sp[p] = advance_bits_fz(g.row_state[ky], 2ULL * ((uint64_t) (a[p] & 65535) * 65536ULL + cx) + 1ULL);
                sm[p] = advance_bits_fz(g.row_state[ky], 2ULL * ((uint64_t) ((-a[p]) & 65535) * 65536ULL + cx) + 1ULL);
            }
        }
        int cur_l = -1, cur_h = -1;
        double Cl[R][4], Ch[R][4];
#pragma unroll 1
        for (int c = 0; c < it.n; c++) {
            const int x = it.x0 + c, kx = x > half ? x - N : x;
            const EigAxis ax = eig_axis(g, eig_index_x(g, kx));
            if (ax.l != cur_l || ax.h != cur_h) {  // (uniform) kx left the table cell
                cur_l = ax.l, cur_h = ax.h;
#pragma unroll
                for (int p = 0; p < R; p++) {
                    // (both x corners whatever this column's fraction is: the next columns of the cell need the upper one)
                    eig_yz_blend(g, ax.l, ay, az[p], Cl[p]);
                    eig_yz_blend(g, ax.h, ay, az[p], Ch[p]);
                }
            }
            const int kxy2 = kx * kx + ky * ky;
            const bool dead_xy = dead_y || (kx < 0 ? -kx : kx) == g.kmax;  // zeldovich.cpp:350
            // sums X, Y, Z, fX, fY, fZ of the two line elements
            double Ar[NJOB], Ai[NJOB], Br[NJOB], Bi[NJOB];
#pragma unroll
            for (int j = 0; j < NJOB; j++) Ar[j] = Ai[j] = Br[j] = Bi[j] = 0.0;
#pragma unroll
            for (int p = 0; p < R; p++) {
                if (p == 1) {
                    // the sums of pair 0 wait in the thread's own two elements of the staged lines while pair 1 is worked on (24 doubles
                    // less to hold beside two Box-Muller chains: without this the kernel spilled 54 registers at 256)
                    __syncthreads();  // the previous column's line waves are done with their exchange areas
#pragma unroll
                    for (int j = 0; j < NJOB; j++) {
                        stage[j * L + kA] = cplx{Ar[j], Ai[j]};
                        stage[j * L + kB] = cplx{Br[j], Bi[j]};
                        Ar[j] = Ai[j] = Br[j] = Bi[j] = 0.0;
                    }
                }
                const int kz = a[p], k2i = kxy2 + kz * kz;
                const bool live = !dead_xy && kz != g.kmax && (g.corner_modes || k2i < g.k2i_cut);
                // the four draws; the generators move on to the next column
                const uint64_t p1 = zdpcg::output(sp[p]);
                u128 s2 = zdpcg::step(sp[p]);
                const uint64_t p2 = zdpcg::output(s2);
                sp[p] = zdpcg::step(s2);
                const uint64_t m1 = zdpcg::output(sm[p]);
                s2 = zdpcg::step(sm[p]);
                const uint64_t m2 = zdpcg::output(s2);
                sm[p] = zdpcg::step(s2);
                if (!__any(live) || ZD_TUNE(S.prune & 8)) continue;  // all 64 pairs of the wave zeroed (bit 3: tuning ablation — draws only)
                // ---- shared by the pair: P(k), 1/k^2, eigenmode, f, rescale ----
                const double k2v = (double) k2i * g.fundamental2;
                const double P   = genf_power<PLAW>(g, T, k2v);
                const double ik2 = frcp(k2v);
                double eh[4];
                {
                    const double wl = 1 - ax.f, wh = ax.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) eh[q] = wl * Cl[p][q];
                    if (wh != 0) {
#pragma unroll
                        for (int q = 0; q < 4; q++) eh[q] = fma(wh, Ch[p][q], eh[q]);
                    }
                }
                // get_eigenmode (zeldovich.cpp:229-276) for kz = +a; the mode -a has e_z negated, everything else the same
                const double n2 = eh[0] * eh[0] + eh[1] * eh[1] + eh[2] * eh[2];
                double rr = trans_rsq(n2);
                rr = rr * fma(-0.5 * n2, rr * rr, 1.5);
                rr = rr * fma(-0.5 * n2, rr * rr, 1.5);
                eh[0] *= rr;
                eh[1] *= rr;
                eh[2] *= rr;
                const double dot = kx * eh[0] + ky * eh[1] + kz * eh[2];
                double norm = (double) k2i * frcp(dot);
                if (!isfinite(norm)) norm = 0.0;
                const double f = (sqrt_pos(1. + 24 * eh[3] * g.f_cluster) - 1) * .25;
                double rescale = 1.0;
                if (g.qPLTrescale) rescale = fexp(g.ln_growth_ratio * (g.target_f - f), T);
                const double sx = rescale * (norm * eh[0]) * g.fundamental * ik2;
                const double sy = rescale * (norm * eh[1]) * g.fundamental * ik2;
                const double sz = rescale * (norm * eh[2]) * g.fundamental * ik2;
                // ---- cgauss<2> of the two modes (power_spectrum.cpp:338-359); zeroed lanes ride along with amplitude 0 ----
                double dpr, dpi, dmr, dmi;
                {
                    const uint64_t u = p1 + 1ULL;  // one_rand<2>: (r + 1) 2^-64, and 1.0 for r = 2^64 - 1 (u = 0)
                    double v = P;
                    if (!g.fixed_power) v = -P * flog(u64_to_double(u), 64, T);
                    v = (u == 0 && !g.fixed_power) || !live ? 0.0 : v;
                    const double amp = sqrt_pos(v);
                    double sn, cs;
                    sincos_u01(u64_to_double(p2 + 1ULL), T, sn, cs);
                    dpr = amp * cs, dpi = amp * sn;
                }
                {
                    const uint64_t u = m1 + 1ULL;
                    double v = P;
                    if (!g.fixed_power) v = -P * flog(u64_to_double(u), 64, T);
                    v = (u == 0 && !g.fixed_power) || !live || (p == 0 && !has_m0) ? 0.0 : v;
                    const double amp = sqrt_pos(v);
                    double sn, cs;
                    sincos_u01(u64_to_double(m2 + 1ULL), T, sn, cs);
                    dmr = amp * cs, dmi = amp * sn;
                }
                vsum = fma(dpr, dpr, fma(dpi, dpi, vsum));
                vsum = fma(dmr, dmr, fma(dmi, dmi, vsum));
                dmr *= fold_m;
                dmi *= fold_m;
                const double fx = f * sx, fy = f * sy, fz = f * sz;
                if (p == 0) {  // "+" -> element kA, "-" -> element kB
                    cmac(Ar[0], Ai[0], sx, dpr, dpi);
                    cmac(Ar[1], Ai[1], sy, dpr, dpi);
                    cmac(Ar[2], Ai[2], sz, dpr, dpi);
                    cmac(Ar[3], Ai[3], fx, dpr, dpi);
                    cmac(Ar[4], Ai[4], fy, dpr, dpi);
                    cmac(Ar[5], Ai[5], fz, dpr, dpi);
                    cmac(Br[0], Bi[0], sx, dmr, dmi);
                    cmac(Br[1], Bi[1], sy, dmr, dmi);
                    cmac(Br[2], Bi[2], -sz, dmr, dmi);
                    cmac(Br[3], Bi[3], fx, dmr, dmi);
                    cmac(Br[4], Bi[4], fy, dmr, dmi);
                    cmac(Br[5], Bi[5], -fz, dmr, dmi);
                } else {  // pair 1 (R = 2): "+" -> kB; "-" -> kA (thread 0: kB as well)
                    cmac(Br[0], Bi[0], sx, dpr, dpi);
                    cmac(Br[1], Bi[1], sy, dpr, dpi);
                    cmac(Br[2], Bi[2], sz, dpr, dpi);
                    cmac(Br[3], Bi[3], fx, dpr, dpi);
                    cmac(Br[4], Bi[4], fy, dpr, dpi);
                    cmac(Br[5], Bi[5], fz, dpr, dpi);
                    const double ar = dmr * to_a, ai = dmi * to_a, br = dmr * to_b, bi = dmi * to_b;
                    cmac(Ar[0], Ai[0], sx, ar, ai);
                    cmac(Ar[1], Ai[1], sy, ar, ai);
                    cmac(Ar[2], Ai[2], -sz, ar, ai);
                    cmac(Ar[3], Ai[3], fx, ar, ai);
                    cmac(Ar[4], Ai[4], fy, ar, ai);
                    cmac(Ar[5], Ai[5], -fz, ar, ai);
                    cmac(Br[0], Bi[0], sx, br, bi);
                    cmac(Br[1], Bi[1], sy, br, bi);
                    cmac(Br[2], Bi[2], -sz, br, bi);
                    cmac(Br[3], Bi[3], fx, br, bi);
                    cmac(Br[4], Bi[4], fy, br, bi);
                    cmac(Br[5], Bi[5], -fz, br, bi);
                }
            }
            // ---- job inputs (zd_kernels.hip genf_tile, GENF_PLTN) times W_N^{k2 residue} -> staged lines ----
            if constexpr (R == 2) {
#pragma unroll
                for (int j = 0; j < NJOB; j++) {
                    const cplx pa = stage[j * L + kA], pb = stage[j * L + kB];
                    Ar[j] += pa.x;
                    Ai[j] += pa.y;
                    Br[j] += pb.x;
                    Bi[j] += pb.y;
                }
            } else {
                __syncthreads();  // the previous column's line waves are done with their exchange areas
            }
            {
                auto put = [&](int j, int k2, double vr, double vi, double wr, double wi) {
                    stage[j * L + k2] = cplx{vr * wr - vi * wi, vr * wi + vi * wr};
                };
                put(0, kA, -Ai[0] - Ar[3], Ar[0] - Ai[3], wAr, wAi);   // JOB_XV_SELF (i - f) s_x D = i X - fX
                put(1, kA, -Ai[0] + Ar[3], Ar[0] + Ai[3], wAr, wAi);   // JOB_XV_TWIN (i + f) s_x D = i X + fX
                put(2, kA, -Ar[2] - Ai[1], -Ai[2] + Ar[1], wAr, wAi);  // JOB_B_SELF  -Z + i Y
                put(3, kA, Ar[2] - Ai[1], Ai[2] + Ar[1], wAr, wAi);    // JOB_B_TWIN   Z + i Y
                put(4, kA, -Ar[5] - Ai[4], -Ai[5] + Ar[4], wAr, wAi);  // JOB_D_SELF  -fZ + i fY
                put(5, kA, Ar[5] - Ai[4], Ai[5] + Ar[4], wAr, wAi);    // JOB_D_TWIN   fZ + i fY
                put(0, kB, -Bi[0] - Br[3], Br[0] - Bi[3], wBr, wBi);
                put(1, kB, -Bi[0] + Br[3], Br[0] + Bi[3], wBr, wBi);
                put(2, kB, -Br[2] - Bi[1], -Bi[2] + Br[1], wBr, wBi);
                put(3, kB, Br[2] - Bi[1], Bi[2] + Br[1], wBr, wBi);
                put(4, kB, -Br[5] - Bi[4], -Bi[5] + Br[4], wBr, wBi);
                put(5, kB, Br[5] - Bi[4], Bi[5] + Br[4], wBr, wBi);
            }
            __syncthreads();
            // ---- waves 0..5: one job line each — transform, Hermitian store ----
            if (wave < NJOB) {
                double re[E], im[E];
                const cplx *line = stage + wave * L;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const cplx v = line[lane + 64 * e];
                    re[e] = v.x;
                    im[e] = v.y;
                }
                wave_lds_sync();  // the line is in registers: its area now serves this wave's exchanges
                if (!ZD_TUNE(S.prune & 128)) fft_line_wave<PL, XL>(re, im, lane, reinterpret_cast<double *>(stage + wave * L), twL);  // bit 7: ablation
                if (ZD_TUNE(S.prune & 16) && re[0] != 123.456) continue;  // bit 4: tuning ablation (no stores)
                const int arr = wave >> 1, twin = wave & 1;
                const int sl = twin ? S.Hq + kyl : kyl;
                const int xs = twin ? ((N - x) & (N - 1)) : x;
                const double sgi = twin ? -1.0 : 1.0;  // twin jobs are stored conjugated
                int l2 = lane;
                asm volatile("" : "+v"(l2));  // keep the address arithmetic behind the transform (registers)
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int z2 = l2 + 64 * e;
                    out[store_elem(S, 0, z2, arr, sl, xs)] = cplx{re[e], sgi * im[e]};
                }
            }
        }
    }
    if (g.accum_var) {  // rows ky >= 1 stand for their Hermitian twins too (k_genf)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
        if (lane == 0) atomicAdd(&g.var_slots[(blockIdx.x * (NT / 64) + wave) % NSLOT], 2.0 * vsum);
    }
}

namespace zd {

bool genz_plt_supported(int N, int L) { return L == 1024 && (N == L || N == 2 * L); }

size_t genz_plt_lds_bytes(const GenConst &g, int L) { return sizeof(double) * (size_t) ((g.genf_n + 3) & ~1) + sizeof(cplx) * 6 * (size_t) L; }

template <int L, int R, bool PLAW>
static int launch_genz_t(const GenConst &g, const StoreLayout &S, int ky0, int residue, unsigned nitems, const FzItem *items,
                         const void *twN, const void *twL, void *out, unsigned *ctr, int ncu, hipStream_t st) {
    const size_t shmem = genz_plt_lds_bytes(g, L);
    set_dyn_lds<k_genz_plt<L, R, PLAW>>(shmem);
    dim3 grid((unsigned) std::min<long long>(nitems, ncu)), block(L / 2);
    hipLaunchKernelGGL((k_genz_plt<L, R, PLAW>), grid, block, shmem, st, g, S, ky0, residue, nitems, items, (const cplx *) twN,
                       (const cplx *) twL, (cplx *) out, ctr);
    ZD_LAUNCH_CHECK();
    return 0;
}

// rows ky0 + S.ky_stride * kyl of the items, z-residue `residue` of a pass with L = N / R points per z line; `ctr`: zeroed counter
int launch_genz_plt(const GenConst &g, const StoreLayout &S, int ky0, int L, int residue, unsigned nitems, const FzItem *items,
                    const void *twN, const void *twL, void *out, unsigned *ctr, int ncu, hipStream_t st) {
    if (!genz_plt_supported(g.N, L) || !g.genf_tab || !g.qPLT || !g.eig) return 2;
    if (nitems == 0) return 0;
    const int R = g.N / L;
#define GZ(l, r)                                                                                                   \
    if (L == l && R == r)                                                                                          \
        return g.is_powerlaw ? launch_genz_t<l, r, true>(g, S, ky0, residue, nitems, items, twN, twL, out, ctr, ncu, st) \
                             : launch_genz_t<l, r, false>(g, S, ky0, residue, nitems, items, twN, twL, out, ctr, ncu, st);
    GZ(1024, 1)
    GZ(1024, 2)
#undef GZ
    return 2;
}

}  // namespace zd
