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

// zdfft::fft_line for a line owned by ONE wave (PL::T == 64) or by half a wave (PL::T == 32: the two halves run the same instructions on
// their own lines and exchange areas): the same passes, the exchanges without workgroup barriers
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

// grid: persistent, one workgroup per CU at L = 1024 (the staged lines + the generator's table image are 118 KB of LDS), two at
// L = 512   block: L / 2
template <int L, int R, bool PLAW>
__global__ __launch_bounds__(L / 2) void k_genz_plt(GenConst g, StoreLayout S, int ky0, int residue, unsigned nitems,
                                                   const FzItem *__restrict__ items, const cplx *__restrict__ twN,
                                                   const cplx *__restrict__ twL, cplx *__restrict__ out,
                                                   unsigned *__restrict__ ctr) {
    static_assert((L == 1024 || L == 512) && (R == 1 || R == 2), "a wave per 1024-point line, half a wave per 512-point line; one or two fold terms");
    constexpr int NT = L / 2, E = 16, NJOB = 6;
    using PL = zdfft::Plan<L, E>;
    using XL = zdfft::LineInner<L, 1>;
    constexpr int TL = PL::T, LPW = 64 / TL;  // threads per line, lines per wave
    static_assert(TL == 64 || TL == 32, "a line is a wave or half a wave");
    extern __shared__ __attribute__((aligned(16))) double T[];  // GenfTab image | work-item slot | staged lines [job][k2]
    for (int i = threadIdx.x; i < g.genf_n / 2; i += NT) reinterpret_cast<double2 *>(T)[i] = reinterpret_cast<const double2 *>(g.genf_tab)[i];
    unsigned *slot = reinterpret_cast<unsigned *>(T + g.genf_n);
    cplx *stage    = reinterpret_cast<cplx *>(T + ((g.genf_n + 3) & ~1));
    // twiddles in LDS: the steady-state loop then has NO vector-memory loads — on gfx9 loads and stores share one in-order counter
    // (vmcnt), so a load issued behind the column's 16 line stores waits for every one of them, and the stores, which should
    // drain behind the next column's arithmetic, were 48 of the stage's 162 ms (profiles/r05_tuning_notes.md).
    //   twl[m] = exp(2 pi i m / L), m < 256: all the inter-pass twiddles a 16 x 16 x 4 (L = 1024) or 16 x 16 x 2 (L = 512) transform asks for
    //   twn[t] = exp(2 pi i t / N), t < L/2: the final factor W_N^{k2 residue} of line element k2 = t (element L - t: -conj)
    constexpr int NTWL = 256;
    cplx *twl = stage + NJOB * L, *twn = twl + NTWL;
    for (int i = threadIdx.x; i < NTWL; i += NT) twl[i] = twL[i];
    for (int i = threadIdx.x; i < L / 2; i += NT) twn[i] = twN[i];
    const int N = g.N, half = g.half;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;

    // ---- what depends on the thread only ----
    // pair p of the thread: |kz| = a[p]; its "+" mode is fold term 0 of line element a[p], its "-" mode fold term R - 1 of element
    // L - a[p].  Line elements owned: kA = t, kB = L - t (thread 0: 0 and L/2)
    int a[R];
    a[0] = t;
    if constexpr (R == 2) a[1] = t == 0 ? L / 2 : L - t;
    const int kA = t, kB = t == 0 ? L / 2 : L - t;
    // final factor W_N^{k2 residue} of the two line elements; fold factor W_R^{(R-1) residue} of the "-" modes
    static_assert(R == 1 || R == 2, "the line twiddles below: residue 0 -> 1; residue 1 at N = 2 L -> twn[t], -conj twn[t]");
    const double fold_m = (R == 2 && (residue & 1)) ? -1.0 : 1.0;
    // thread 0: pair 0 is the single mode kz = 0; at R = 2 its pair 1 (kz = +-L/2) puts BOTH modes on element kB
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
                    const EigAxis az = eig_axis(g, a[p]);  // the table's +kz half space
                    eig_yz_blend(g, ax.l, ay, az, Cl[p]);
                    eig_yz_blend(g, ax.h, ay, az, Ch[p]);
                }
            }
            const int kxy2 = kx * kx + ky * ky;
            const bool dead_xy = dead_y || (kx < 0 ? -kx : kx) == g.kmax;  // zeldovich.cpp:350
            // W_N^{k2 residue} of the two line elements: twn[t] for kA = t; kB = L - t: exp(2 pi i (L - t) / N) = -conj twn[t] (N = 2 L);
            // thread 0: kB = L/2 -> i.  residue 0 (and R = 1): 1
            double wAr = 1.0, wAi = 0.0, wBr = 1.0, wBi = 0.0;
            if (R == 2 && (residue & 1)) {
                int tl = t;
                asm volatile("" : "+v"(tl));  // (address formed here: see kzl below)
                const cplx wa = twn[tl];
                wAr = wa.x, wAi = wa.y;
                wBr = t == 0 ? 0.0 : -wa.x, wBi = t == 0 ? 1.0 : wa.y;
            }
            // The six job inputs of ONE mode (zd_kernels.hip genf_tile, GENF_PLTN: i X - fX, i X + fX, -Z + i Y, Z + i Y, -fZ + i fY,
            // fZ + i fY with X = s_x D ...) times the line twiddle w: linear in the mode, so every mode goes straight into its line
            // element in LDS — pair 0 writes the thread's two elements, pair 1 adds to them — and no sums are held in registers
            // beside the Box-Muller chains (24 doubles: with them the kernel spilled, and spilled values come back through
            // scratch LOADS, which wait behind the line stores in the vmcnt queue)
            auto mode_jobs = [&](double dr, double di, double wr, double wi, double sx_, double sy_, double sz_, double f_,
                                 double (&vr)[NJOB], double (&vi)[NJOB]) {
                const double er = dr * wr - di * wi, ei = dr * wi + di * wr;
                const double p1 = sx_ * er, p2 = sx_ * ei, q1 = f_ * p1, q2 = f_ * p2;
                vr[0] = -q1 - p2, vi[0] = p1 - q2;  // JOB_XV_SELF (i - f) s_x D
                vr[1] = q1 - p2, vi[1] = p1 + q2;   // JOB_XV_TWIN (i + f) s_x D
                const double r1 = sz_ * er, r2 = sz_ * ei, t1 = sy_ * er, t2 = sy_ * ei;
                vr[2] = -r1 - t2, vi[2] = t1 - r2;  // JOB_B_SELF  (-s_z + i s_y) D
                vr[3] = r1 - t2, vi[3] = t1 + r2;   // JOB_B_TWIN  ( s_z + i s_y) D
                const double u1 = f_ * r1, u2 = f_ * r2, v1 = f_ * t1, v2 = f_ * t2;
                vr[4] = -u1 - v2, vi[4] = v1 - u2;  // JOB_D_SELF  f (-s_z + i s_y) D
                vr[5] = u1 - v2, vi[5] = v1 + u2;   // JOB_D_TWIN  f ( s_z + i s_y) D
            };
#pragma unroll
            for (int p = 0; p < R; p++) {
                const int kz = a[p], k2i = kxy2 + kz * kz;
                const bool live = !dead_xy && kz != g.kmax && (g.corner_modes || k2i < g.k2i_cut);
                // ZD_qonemode (zeldovich.cpp:354-356): only the mode one_mode survives — the two modes of a pair then differ
                const bool om = g.qonemode != 0, omxy = kx == g.one_mode[0] && ky == g.one_mode[1];
                const bool liveP = live && (!om || (omxy && kz == g.one_mode[2]));
                const bool liveM = live && (!om || (omxy && -kz == g.one_mode[2]));
                // the four draws; the generators move on to the next column
                const uint64_t p1 = zdpcg::output(sp[p]);
                u128 s2 = zdpcg::step(sp[p]);
                const uint64_t p2 = zdpcg::output(s2);
                sp[p] = zdpcg::step(s2);
                const uint64_t m1 = zdpcg::output(sm[p]);
                s2 = zdpcg::step(sm[p]);
                const uint64_t m2 = zdpcg::output(s2);
                sm[p] = zdpcg::step(s2);
                double vPr[NJOB], vPi[NJOB], vMr[NJOB], vMi[NJOB];
                const bool any = __any(liveP || liveM) && !ZD_TUNE(S.prune & 8);  // all 64 pairs of the wave zeroed? (bit 3: tuning ablation — draws only)
                if (any) {
                    // ---- shared by the pair: P(k), eigenmode, f, rescale ----
                    const double k2v = (double) k2i * g.fundamental2;
                    const double P   = genf_power<PLAW>(g, T, k2v);
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
                    // get_eigenmode (zeldovich.cpp:229-276) for kz = +a; the mode -a has e_z negated, everything else the same.  The
                    // reference normalises e, forms e k^2 / (k.e) and the generator multiplies by fundamental / k^2: the coefficient
                    // s_j = e_j / ((k.e) fundamental) does not depend on |e| or on k^2, so neither is computed (one reciprocal instead of
                    // an inverse square root with two Newton steps and two reciprocals); k.e = 0 or not finite -> 0 as there (:262-263)
                    int kzl = kz;
                    asm volatile("" : "+v"(kzl));  // (double) kz is converted here, per column: hoisted out of the loop it was spilled,
                                                   // and its reload is a vector-memory load in the hot loop (see twl above)
                    const double dot = kx * eh[0] + ky * eh[1] + kzl * eh[2];
                    double inv = frcp(dot * g.fundamental);
                    if (!isfinite(inv)) inv = 0.0;
                    const double f = (sqrt_pos(1. + 24 * eh[3] * g.f_cluster) - 1) * .25;
                    if (g.qPLTrescale) inv *= fexp(g.ln_growth_ratio * (g.target_f - f), T);
                    const double sx = inv * eh[0], sy = inv * eh[1], sz = inv * eh[2];
                    // ---- cgauss<2> of the two modes (power_spectrum.cpp:338-359); zeroed lanes ride along with amplitude 0 ----
                    double dpr, dpi, dmr, dmi;
                    {
                        const uint64_t u = p1 + 1ULL;  // one_rand<2>: (r + 1) 2^-64, and 1.0 for r = 2^64 - 1 (u = 0)
                        double v = P;
                        if (!g.fixed_power) v = -P * flog(u64_to_double(u), 64, T);
                        v = (u == 0 && !g.fixed_power) || !liveP ? 0.0 : v;
                        const double amp = sqrt_pos(v);
                        double sn, cs;
                        sincos_u01(u64_to_double(p2 + 1ULL), T, sn, cs);
                        dpr = amp * cs, dpi = amp * sn;
                    }
                    {
                        const uint64_t u = m1 + 1ULL;
                        double v = P;
                        if (!g.fixed_power) v = -P * flog(u64_to_double(u), 64, T);
                        v = (u == 0 && !g.fixed_power) || !liveM || (p == 0 && !has_m0) ? 0.0 : v;
                        const double amp = sqrt_pos(v);
                        double sn, cs;
                        sincos_u01(u64_to_double(m2 + 1ULL), T, sn, cs);
                        dmr = amp * cs, dmi = amp * sn;
                    }
                    vsum = fma(dpr, dpr, fma(dpi, dpi, vsum));
                    vsum = fma(dmr, dmr, fma(dmi, dmi, vsum));
                    dmr *= fold_m;
                    dmi *= fold_m;
                    // pair 0: "+" -> element kA, "-" -> kB.   pair 1 (R = 2): "+" -> kB, "-" -> kA (thread 0: kB as well)
                    const bool m_to_a = p == 1 && has_m0;
                    mode_jobs(dpr, dpi, p == 0 ? wAr : wBr, p == 0 ? wAi : wBi, sx, sy, sz, f, vPr, vPi);
                    mode_jobs(dmr, dmi, m_to_a ? wAr : wBr, m_to_a ? wAi : wBi, sx, sy, -sz, f, vMr, vMi);
                } else if (p == 0) {
#pragma unroll
                    for (int j = 0; j < NJOB; j++) vPr[j] = vPi[j] = vMr[j] = vMi[j] = 0.0;
                }
                if (p == 0) {
                    __syncthreads();  // the previous column's line waves are done with their exchange areas
#pragma unroll
                    for (int j = 0; j < NJOB; j++) {
                        stage[j * L + kA] = cplx{vPr[j], vPi[j]};
                        stage[j * L + kB] = cplx{vMr[j], vMi[j]};
                    }
                } else if (any) {
                    const int km = has_m0 ? kA : kB;
#pragma unroll
                    for (int j = 0; j < NJOB; j++) {
                        cplx q = stage[j * L + kB];
                        q.x += vPr[j], q.y += vPi[j];
                        stage[j * L + kB] = q;
                    }
#pragma unroll
                    for (int j = 0; j < NJOB; j++) {  // (behind the loop above: thread 0 adds both modes to the same element)
                        cplx q = stage[j * L + km];
                        q.x += vMr[j], q.y += vMi[j];
                        stage[j * L + km] = q;
                    }
                }
            }
            __syncthreads();
            // ---- the first 6 / LPW waves: one job line per wave (L = 1024) or per half wave (L = 512) — transform, Hermitian store ----
            if (wave < NJOB / LPW) {
                double re[E], im[E];
                const int jl = wave * LPW + lane / TL, tl = lane % TL;  // job line, thread of the line
                const cplx *line = stage + jl * L;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const cplx v = line[tl + TL * e];
                    re[e] = v.x;
                    im[e] = v.y;
                }
                wave_lds_sync();  // the line is in registers: its area now serves its threads' exchanges
                int l3 = tl;
                asm volatile("" : "+v"(l3));  // (the transform's LDS addresses are formed per column, not carried — and spilled — across the loop)
                if (!ZD_TUNE(S.prune & 128)) fft_line_wave<PL, XL>(re, im, l3, reinterpret_cast<double *>(stage + jl * L), twl);  // bit 7: ablation
                if (ZD_TUNE(S.prune & 16) && re[0] != 123.456) continue;  // bit 4: tuning ablation (no stores)
                const int arr = jl >> 1, twin = jl & 1;
                const int sl = twin ? S.Hq + kyl : kyl;
                const int xs = twin ? ((N - x) & (N - 1)) : x;
                const double sgi = twin ? -1.0 : 1.0;  // twin jobs are stored conjugated
                int l2 = tl;
                asm volatile("" : "+v"(l2));  // keep the address arithmetic behind the transform (registers)
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int z2 = l2 + TL * e;
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

bool genz_plt_supported(int N, int L) { return (L == 1024 || L == 512) && (N == L || N == 2 * L); }

size_t genz_plt_lds_bytes(const GenConst &g, int L) {  // table image + slot | 6 staged lines | twl[L/4] | twn[L/2]
    return sizeof(double) * (size_t) ((g.genf_n + 3) & ~1) + sizeof(cplx) * (6 * (size_t) L + 256 + L / 2);
}

template <int L, int R, bool PLAW>
static int launch_genz_t(const GenConst &g, const StoreLayout &S, int ky0, int residue, unsigned nitems, const FzItem *items,
                         const void *twN, const void *twL, void *out, unsigned *ctr, int ncu, hipStream_t st) {
    const size_t shmem = genz_plt_lds_bytes(g, L);
    set_dyn_lds<k_genz_plt<L, R, PLAW>>(shmem);
    dim3 grid((unsigned) std::min<long long>(nitems, (long long) ncu * (L == 512 ? 2 : 1))), block(L / 2);
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
    GZ(512, 1)
    GZ(512, 2)
#undef GZ
    return 2;
}

}  // namespace zd
