// zd_kernels_np2.hip — kernels for PPD = 2^a 3^b (not a power of two): the composite-length line transform of zd_fft_q.h
// in the roles of k_zfft_f / k_yfft_f / k_xfft_seq (field store, ZA, one rank).  A separate translation unit so that the
// power-of-two kernels (zd_kernels.hip) and their build time are untouched.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "zd_device.h"
#include "zd_epi.h"
#include "zd_fft_q.h"
#include "zd_launch.h"

using namespace zd;
using zdfft::cplx;


#ifdef ZD_TESTING
// ------------------------------------------------------------------------------------------------
// test kernels: batches of independent lines of length P*Q through the two LDS layouts
template <int P, int E, int Q, int W>
__global__ __launch_bounds__(W *Q *P / E) void k_test_fftq_cols(const cplx *__restrict__ twP, const cplx *__restrict__ twN,
                                                              const cplx *__restrict__ twQ, const cplx *__restrict__ in,
                                                              cplx *__restrict__ out, long long lines) {
    // data layout [n][lines] (line index contiguous): the strided-axis situation of the y/z passes
    using LQ = zdfft::LineQ<P, E, Q, W, false>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T;
    const int c = threadIdx.x % (W * Q), t = threadIdx.x / (W * Q);
    const int w = c % W, n2 = c / W;
    const long long col = (long long) blockIdx.x * W + w;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = in[(long long) (Q * (t + T * e) + n2) * lines + col];
        re[e] = v.x;
        im[e] = v.y;
    }
    LQ::run(re, im, t, w, n2, lds, twP, twN, twQ);
#pragma unroll
    for (int e = 0; e < E; e++) out[(long long) ((t + T * e) + P * n2) * lines + col] = cplx{re[e], im[e]};
}
template <int P, int E, int Q, int W>
__global__ __launch_bounds__(W *Q *P / E) void k_test_fftq_lines(const cplx *__restrict__ twP, const cplx *__restrict__ twN,
                                                               const cplx *__restrict__ twQ, const cplx *__restrict__ in,
                                                               cplx *__restrict__ out, long long lines) {
    // data layout [lines][n]: contiguous lines (x pass)
    using LQ = zdfft::LineQ<P, E, Q, W, true>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T, N = P * Q;
    const int t = threadIdx.x % T, c = threadIdx.x / T;
    const int w = c % W, n2 = c / W;
    const long long line = (long long) blockIdx.x * W + w;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = in[line * N + Q * (t + T * e) + n2];
        re[e] = v.x;
        im[e] = v.y;
    }
    LQ::run(re, im, t, w, n2, lds, twP, twN, twQ);
#pragma unroll
    for (int e = 0; e < E; e++) out[line * N + (t + T * e) + P * n2] = cplx{re[e], im[e]};
}

#endif  // ZD_TESTING
// ------------------------------------------------------------------------------------------------
// Field-store pipeline for PPD = P*Q (ZA, one rank): the roles of k_zfft_f, k_yfft_f and k_xfft_seq of zd_kernels.hip with
// the composite line transform.  A thread enters a transform with elements Q (t + T e) + n2 of its line and leaves with
// elements (t + T e) + P n2: the load stage of every kernel uses the first index, the store stage the second.

// z stage: lines of length L = P*Q (the folded z direction) for NC columns x the 8 rows of a row group.
//   grid: (N/NC, row groups, nfield)   block: NC*8*Q*P/E
template <int P, int E, int Q, int NC>
// (held to 128 registers — 11-18 spilled dwords for Q = 27 — so that two of its 7-wave workgroups fit a CU: Z stage of PPD=3456
// 520 -> 491 ms, of PPD=6912 k_cutoff=2 1244 -> 1155)
__global__ __launch_bounds__(NC *FIELD_RB *Q *P / E, 4) void k_zfft_fq(FieldLayout F, StoreLayout S, int ky0, int kyloc0, int nky,
                                                                  const cplx *__restrict__ Y, const cplx *__restrict__ twP,
                                                                  const cplx *__restrict__ twL, const cplx *__restrict__ twQ,
                                                                  cplx *__restrict__ out) {
    constexpr int W = NC * FIELD_RB, L = P * Q;
    using LQ = zdfft::LineQ<P, E, Q, W, false>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T;
    const int N = S.N;
    const int c = threadIdx.x % (W * Q), t = threadIdx.x / (W * Q);
    const int w = c % W, n2 = c / W;
    const int grp = blockIdx.y;
    const int r = w & (FIELD_RB - 1), x = blockIdx.x * NC + w / FIELD_RB;
    const int ky = ky0 + (grp * FIELD_RB + r) * S.ky_stride;
    if (S.prune & 2) {
        if (__syncthreads_and(column_is_zero(S, x > S.half ? x - N : x, ky))) return;
    }
    const cplx *src = Y + ((((long long) blockIdx.z * (nky / FIELD_RB) + grp) * L) * N) * FIELD_RB;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int k2 = Q * (t + T * e) + n2;
        const cplx v = src[((long long) k2 * N + x) * FIELD_RB + r];
        re[e] = v.x;
        im[e] = v.y;
    }
    LQ::run(re, im, t, w, n2, lds, twP, twL, twQ);
    const FieldRow row = F.rows[kyloc0 / FIELD_RB + grp];
    const unsigned pos = ((unsigned) (x < row.split ? x : x - row.gap)) * FIELD_RB + r;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int z2 = (t + T * e) + P * n2, dst = z2 / F.Zq, zl = z2 - dst * F.Zq;  // destination rank, its local plane
        out[(long long) dst * F.chunk_elems + (long long) (zl * F.nfield + (int) blockIdx.z) * F.field_elems + (unsigned) row.base + pos]
            = cplx{re[e], im[e]};
    }
}

// y stage: builds (qx + i qy)_r0 | (qx + i qy)_r1 | qz_r0 + i qz_r1 from the potentials (see k_yfft_f) and transforms
// along y; ring rows are indexed by y directly.
//   grid: (3*N/W, 1, planes)   block: W*Q*P/E
template <int P, int E, int Q, int W>
__global__ __launch_bounds__(W *Q *P / E) void k_yfft_fq(FieldLayout F, StoreLayout S, const cplx *__restrict__ twP,
                                                        const cplx *__restrict__ twN, const cplx *__restrict__ twQ,
                                                        const cplx *__restrict__ store, int plane0, int ring_pitch,
                                                        cplx *__restrict__ ring, int dens) {
    constexpr int N = P * Q, NT = N / W;
    using LQ = zdfft::LineQ<P, E, Q, W, false>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T;
    const int c = threadIdx.x % (W * Q), t = threadIdx.x / (W * Q);
    const int w = c % W, n2 = c / W;
    int tile, a;
    ytile_of<NT, W>((int) blockIdx.x, tile, a);  // XCD-aware order (zd_device.h)
    // dens (ZD_qdensity = 1, six-field store): a launch of its own over the NT column tiles builds the density array
    //   delta_r0 + i delta_r1 = D_0 + i D_1   (rows y > N/2: conj D_0 + i conj D_1 — D is Hermitian like E)
    // from the fields 4, 5 into `ring` = the density ring [plane][y][x]
    if (dens) {
        tile = (int) blockIdx.x;
        a    = 0;
    }
    const int x = tile * W + w, xm = x ? N - x : 0;
    const int zl = plane0 + blockIdx.z;
    const int kx = x > N / 2 ? x - N : x;
    // a tile without a live row: see k_yfft_f.  Not for the density array: k_xdens_q reads every column of its ring (it has no
    // xdead_lo / hi), so its dead tiles must arrive as zeros — which the `skip` of every row below makes them, without a load
    if (!dens && (S.prune & PRUNE_YTILE) && __syncthreads_and(column_is_zero(S, kx, 0))) return;
    // ZA: E_a alone (a < 2) or (Z_0, Z_1); PLT: the pairs (X, fX), (Y, Z), (fY, fZ) of the six sums — see k_yfft_f
    const bool plt = F.nfield == 6 && !F.ndens, two = plt || a == 2 || dens;
    const int f0 = dens ? 4 : (plt ? (a == 0 ? 0 : (a == 1 ? 1 : 4)) : (a == 2 ? 1 : 2 * a));
    const int f1 = dens ? 5 : (plt ? (a == 0 ? 3 : (a == 1 ? 2 : 5)) : 3);
    const cplx *p0 = store + (long long) (zl * F.nfield + f0) * F.field_elems;
    const long long d01 = two ? (long long) (f1 - f0) * F.field_elems : 0;
    double re[E], im[E];
    FieldRow rows[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int y = Q * (t + T * e) + n2;
        int kyp = y > N / 2 ? N - y : y;
        kyp = kyp < N / 2 ? kyp : N / 2 - 1;
        rows[e] = F.rows[(kyp >> F.lG) / FIELD_RB];  // row slot kyp / G of the chunk of rank kyp % G
    }
    const int gmask = (1 << F.lG) - 1;
    constexpr int BATCH = E >= 4 ? 4 : E;
#pragma unroll
    for (int b = 0; b < E; b += BATCH) {
        cplx u[BATCH], v[BATCH];
        bool skip[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int e = b + j, y = Q * (t + T * e) + n2;
            const bool mir = y > N / 2;
            const int kyp = mir ? N - y : y;
            const int xs  = mir ? xm : x;
            skip[j] = (2 * y == N) || ((S.prune & 4) && column_is_zero(S, kx, kyp));
            const int split = rows[e].split, gap = rows[e].gap;
            const unsigned off = skip[j] ? 0u : (unsigned) rows[e].base + (unsigned) (xs < split ? xs : xs - gap) * FIELD_RB
                                                    + (unsigned) ((kyp >> F.lG) & (FIELD_RB - 1));
            const cplx *q = p0 + ((long long) (kyp & gmask) * F.chunk_elems + off);
            u[j] = q[0];
            if (two) v[j] = q[d01];
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int e = b + j, y = Q * (t + T * e) + n2;
            const bool mir = y > N / 2;
            const double s = mir ? -1.0 : 1.0;
            double vr, vi;
            if (dens) {
                vr = u[j].x - s * v[j].y;
                vi = s * u[j].y + v[j].x;
            } else if (two) {
                vr = -u[j].y - s * v[j].x;
                vi = s * u[j].x - v[j].y;
            } else {
                const double dky = (double) (mir ? N - y : y), dkx = (double) kx;
                vr = -s * (dky * u[j].x + dkx * u[j].y);
                vi = dkx * u[j].x - dky * u[j].y;
            }
            re[e] = skip[j] ? 0.0 : vr;
            im[e] = skip[j] ? 0.0 : vi;
        }
    }
    LQ::run(re, im, t, w, n2, lds, twP, twN, twQ);
    cplx *base = ring + ((long long) (dens ? (int) blockIdx.z : (int) blockIdx.z * 3 + a) * N) * ring_pitch + x;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int y = (t + T * e) + P * n2;
        base[(long long) y * ring_pitch] = cplx{re[e], im[e]};
    }
}

// x stage + epilogue, three lines per workgroup (PPD <= 5461: 3 N / 16 threads): the arrays of one row side by side; the
// threads of lines 0 / 1 hold qx, qy of their plane, only qz_r0 + i qz_r1 (line 2) goes through LDS, records leave straight
// from the registers (the form of k_xfft's field-store path).
//   grid: (N, planes)   block: 3*Q*P/E
template <int P, int E, int Q, bool PLT>
__global__ __launch_bounds__(3 * Q *P / E) void k_xfft_q3(EpiConst ec, const cplx *__restrict__ twP, const cplx *__restrict__ twN,
                                                        const cplx *__restrict__ twQ, const cplx *__restrict__ ring,
                                                        int ring_pitch, int z_first, int z_step, char *__restrict__ records,
                                                        Reduce *__restrict__ red) {
    constexpr int N = P * Q;
    using LQ = zdfft::LineQ<P, E, Q, 3, true>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T, NT = 3 * T * Q;
    const int t = threadIdx.x % T, c = threadIdx.x / T;
    const int a = c % 3, n2 = c / 3;
    const int y = blockIdx.x, pl = blockIdx.y;
    const cplx *src = ring + ((long long) (pl * 3 + a) * N + y) * ring_pitch;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int xi = Q * (t + T * e) + n2;
        const bool dead = x_is_dead(ec, xi);  // (columns of the ring the y stage does not write: zd_device.h EpiConst)
        cplx v = src[dead ? 0 : xi];
        if (dead) v = cplx{0.0, 0.0};
        re[e] = v.x;
        im[e] = v.y;
    }
    LQ::run(re, im, t, a, n2, lds, twP, twN, twQ);
    const int z = z_first + z_step * (int) blockIdx.y;
    MaxAbs mx;
    const long long plane_rec0 = 2 * (long long) blockIdx.y * N * N;
    double2 *cz = reinterpret_cast<double2 *>(lds);  // [x] = {qz_r0, qz_r1}
    if constexpr (PLT) {
        // PLT ring qx + i vx | qy + i qz | vy + i vz: ONE plane per store plane; lines 0 and 2 go through LDS ([x] and
        // [N + x]), line 1 writes the records
        if (a != 1) {
#pragma unroll
            for (int e = 0; e < E; e++) cz[(a ? N : 0) + (t + T * e) + P * n2] = double2{re[e], im[e]};
        }
        __syncthreads();
        if (a == 1) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int xx = (t + T * e) + P * n2;
                const double2 c0 = cz[xx], c2 = cz[N + xx];
                const double pos[3] = {c0.x, re[e], im[e]};
                const double vel[3] = {c0.y * ec.vnorm, c2.x * ec.vnorm, c2.y * ec.vnorm};
                max_track(mx, pos, ((unsigned long long) z * N + (unsigned long long) y) * N + (unsigned long long) xx);
                if (records) emit_record(records, (long long) blockIdx.y * N * N + (long long) y * N + xx, ec, z, y, xx, pos, vel);
            }
        }
    } else {
    if (a == 2) {
#pragma unroll
        for (int e = 0; e < E; e++) cz[(t + T * e) + P * n2] = double2{re[e], im[e]};
    }
    __syncthreads();
    if (a < 2) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xx = (t + T * e) + P * n2;
            const double2 cv = cz[xx];
            const double pos[3] = {re[e], im[e], a ? cv.y : cv.x};
            const double vel[3] = {pos[0] * ec.vnorm, pos[1] * ec.vnorm, pos[2] * ec.vnorm};
            max_track(mx, pos, ((unsigned long long) (z + a * ec.z_pair) * N + (unsigned long long) y) * N + (unsigned long long) xx);
            if (records) emit_record(records, plane_rec0 + (long long) a * N * N + (long long) y * N + xx, ec, z + a * ec.z_pair, y, xx, pos, vel);
        }
    }
    }
    // workgroup reduction through LDS (the workgroup is N/16 threads per line: not a whole number of waves, so no wave shuffles)
    max_reduce_lds<NT>(lds, red, mx);
}

// x stage + epilogue where the three lines of a row do not fit one workgroup (PPD > 5461): one line per workgroup, two
// launches.  Launch 0 (grid (N, planes, 1)) transforms qz_r0 + i qz_r1 (array 2) over its own ring row; launch 1 (grid
// (N, planes, 2)) transforms (qx + i qy)_r, r = blockIdx.z, and writes the records of its plane straight from the
// registers, reading qz from the row launch 0 left (one extra ring-row round trip instead of holding a second line's
// results in registers: that cost 240-450 spilled dwords in every single-launch form tried).
//   block: Q*P/E
template <int P, int E, int Q, bool PLT>
__global__ __launch_bounds__(Q *P / E, 4) void k_xfft_seq_q(EpiConst ec, const cplx *__restrict__ twP, const cplx *__restrict__ twN,
                                                       const cplx *__restrict__ twQ, cplx *ring, int emit,
                                                       int ring_pitch, int z_first, int z_step, char *__restrict__ records,
                                                       Reduce *__restrict__ red) {
    constexpr int N = P * Q;
    using LQ = zdfft::LineQ<P, E, Q, 1, true>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T, NT = T * Q;
    const int t = threadIdx.x % T, n2 = threadIdx.x / T;
    const int y = blockIdx.x, pl = blockIdx.y;
    // ZA ring: launch 0 = array 2, launch 1 = arrays 0, 1 (the two planes).  PLT ring qx + i vx | qy + i qz | vy + i vz:
    // launch 0 = arrays 0 and 2 (grid z = 2), launch 1 = array 1 with the records of the one plane.
    cplx *czrow = ring + ((long long) (pl * 3 + 2) * N + y) * ring_pitch;
    cplx *row0  = ring + ((long long) (pl * 3 + 0) * N + y) * ring_pitch;
    const int a = PLT ? (emit ? 1 : 2 * (int) blockIdx.z) : (emit ? (int) blockIdx.z : 2);
    cplx *src = ring + ((long long) (pl * 3 + a) * N + y) * ring_pitch;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int xi = Q * (t + T * e) + n2;
        const bool dead = x_is_dead(ec, xi);  // (columns of the ring the y stage does not write: zd_device.h EpiConst)
        cplx v = src[dead ? 0 : xi];
        if (dead) v = cplx{0.0, 0.0};
        re[e] = v.x;
        im[e] = v.y;
    }
    LQ::run(re, im, t, 0, n2, lds, twP, twN, twQ);
    if (!emit) {
#pragma unroll
        for (int e = 0; e < E; e++) src[(t + T * e) + P * n2] = cplx{re[e], im[e]};
        // max_disp of the displacement component(s) this launch holds in registers (see k_xfft_two): PLT array 0 = qx + i vx ->
        // axis 0; ZA array 2 = qz_r0 + i qz_r1 -> axis 2 of the two planes of the pair
        if (!PLT || a == 0) {
            constexpr int J = PLT ? 0 : 2;
            double v0 = 0.0, v1 = 0.0;
            int t0 = 0, t1 = 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int xx = (t + T * e) + P * n2;
                const bool g0 = fabs(re[e]) > fabs(v0);
                v0 = g0 ? re[e] : v0;
                t0 = g0 ? xx : t0;
                if constexpr (!PLT) {
                    const bool g1 = fabs(im[e]) > fabs(v1);
                    v1 = g1 ? im[e] : v1;
                    t1 = g1 ? xx : t1;
                }
            }
            const int z0 = z_first + z_step * (int) blockIdx.y;
            MaxAbs mg;
            mg.v[J]   = v0;
            mg.lin[J] = ((unsigned long long) z0 * N + (unsigned long long) y) * N + (unsigned) t0;
            if constexpr (!PLT) {
                const unsigned long long l1 = ((unsigned long long) (z0 + ec.z_pair) * N + (unsigned long long) y) * N + (unsigned) t1;
                if (max_better(v1, l1, mg.v[J], mg.lin[J])) {
                    mg.v[J]   = v1;
                    mg.lin[J] = l1;
                }
            }
            max_reduce_lds<NT>(lds, red, mg);
        }
        return;
    }
    const int z = z_first + z_step * (int) blockIdx.y + (PLT ? 0 : a * ec.z_pair);
    MaxAbs mx;
    const long long rec0 = PLT ? (long long) blockIdx.y * N * N + (long long) y * N
                               : 2 * (long long) blockIdx.y * N * N + (long long) a * N * N + (long long) y * N;
    // max_disp: the two components held in registers, in a loop of their own in front of the records (the third one: launch 0)
    {
        constexpr int JA = PLT ? 1 : 0, JB = PLT ? 2 : 1;
        const unsigned long long rowbase = ((unsigned long long) z * N + (unsigned long long) y) * N;
        int ta = 0, tb = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xx = (t + T * e) + P * n2;
            const bool ga = fabs(re[e]) > fabs(mx.v[JA]), gb = fabs(im[e]) > fabs(mx.v[JB]);
            mx.v[JA] = ga ? re[e] : mx.v[JA];
            ta       = ga ? xx : ta;
            mx.v[JB] = gb ? im[e] : mx.v[JB];
            tb       = gb ? xx : tb;
        }
        mx.lin[JA] = rowbase + (unsigned) ta;
        mx.lin[JB] = rowbase + (unsigned) tb;
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int xx = (t + T * e) + P * n2;
        const cplx c2 = czrow[xx];
        double pos[3], vel[3];
        if constexpr (PLT) {
            const cplx c0 = row0[xx];
            pos[0] = c0.x; pos[1] = re[e]; pos[2] = im[e];
            vel[0] = c0.y * ec.vnorm; vel[1] = c2.x * ec.vnorm; vel[2] = c2.y * ec.vnorm;
        } else {
            pos[0] = re[e]; pos[1] = im[e]; pos[2] = a ? c2.y : c2.x;
            vel[0] = pos[0] * ec.vnorm; vel[1] = pos[1] * ec.vnorm; vel[2] = pos[2] * ec.vnorm;
        }
        if (records) emit_record(records, rec0 + xx, ec, z, y, xx, pos, vel);
    }
    // workgroup reduction through LDS (the workgroup is N/16 threads per line: not a whole number of waves, so no wave shuffles)
    max_reduce_lds<NT>(lds, red, mx);
}

// The ZA ring in ONE launch where three lines do not fit a workgroup (the form k_xfft_seq has for PPD = 8192): Q*P/E threads own
// a row and transform its arrays one after the other.  Array 2 = qz_r0 + i qz_r1 goes first: its real part (plane r0, written
// first) stays in registers, its imaginary part waits in LDS behind the transform's area; then each of (qx + i qy)_r is
// transformed and its plane's records leave straight from the registers.  Against the two-launch form: no round trip of array 2
// through the ring (32 of 88 bytes per particle at PPD = 6912).
//   grid: (N, planes)   block: Q*P/E
template <int P, int E, int Q>
__global__ __launch_bounds__(Q *P / E) void k_xfft_seq1_q(EpiConst ec, const cplx *__restrict__ twP, const cplx *__restrict__ twN,
                                                        const cplx *__restrict__ twQ, const cplx *__restrict__ ring, int ring_pitch,
                                                        int z_first, int z_step, char *__restrict__ records,
                                                        Reduce *__restrict__ red) {
    constexpr int N = P * Q;
    using LQ = zdfft::LineQ<P, E, Q, 1, true>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T, NT = T * Q;
    const int t = threadIdx.x % T, n2 = threadIdx.x / T;
    const int y = blockIdx.x, pl = blockIdx.y;
    double *czi = lds + LQ::LDS_DOUBLES;  // [x]: Im of array 2; a thread reads back only what it wrote itself
    double cr[E];
#pragma unroll
    for (int e = 0; e < E; e++) cr[e] = 0.0;
    const int z = z_first + z_step * (int) blockIdx.y;
    MaxAbs mx;
    // ONE copy of the transform in a rolled loop over the arrays 2, 0, 1 (three inlined copies spilled 350-760 registers)
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        const int a = it == 0 ? 2 : it - 1;
        const cplx *src = ring + ((long long) (pl * 3 + a) * N + y) * ring_pitch;
        double re[E], im[E];
        int ta = t;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xi = Q * (ta + T * e) + n2;
            const bool dead = x_is_dead(ec, xi);  // (columns of the ring the y stage does not write: zd_device.h EpiConst)
            cplx v = src[dead ? 0 : xi];
            if (dead) v = cplx{0.0, 0.0};
            re[e] = v.x;
            im[e] = v.y;
        }
        __syncthreads();  // the previous transform's last LDS reads are done
        LQ::run(re, im, t, 0, n2, lds, twP, twN, twQ);
        int t2 = t;
        asm volatile("" : "+v"(t2));
        if (it == 0) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                cr[e] = re[e];
                czi[(t2 + T * e) + P * n2] = im[e];
            }
            continue;
        }
        const long long rec0 = 2 * (long long) blockIdx.y * N * N + (long long) a * N * N + (long long) y * N;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xx = (t2 + T * e) + P * n2;
            const double pos[3] = {re[e], im[e], a ? czi[xx] : cr[e]};
            const double vel[3] = {pos[0] * ec.vnorm, pos[1] * ec.vnorm, pos[2] * ec.vnorm};
            max_track(mx, pos, ((unsigned long long) (z + a * ec.z_pair) * N + (unsigned long long) y) * N + (unsigned long long) xx);
            if (records) emit_record(records, rec0 + xx, ec, z + a * ec.z_pair, y, xx, pos, vel);
        }
    }
    // workgroup reduction through LDS (the workgroup is N/16 threads per line: not a whole number of waves, so no wave shuffles)
    max_reduce_lds<NT>(lds, red, mx);
}

// x stage of the density array (ZD_qdensity = 1 on the six-field store): one row of delta_r0 + i delta_r1 per workgroup -> the
// float32 density planes of the two residues (WriteParticlesSlab's dens_out, src/output.cpp:93-101,217-224).  density_variance
// comes from the generator (sum |D|^2), as for every packed store.
//   grid: (N, planes)   block: Q*P/E       density: [2 * planes][N][N] in delivery order (r0 plane, r1 plane per store plane)
template <int P, int E, int Q>
__global__ __launch_bounds__(Q *P / E) void k_xdens_q(const cplx *__restrict__ twP, const cplx *__restrict__ twN, const cplx *__restrict__ twQ,
                                                    const cplx *__restrict__ dring, int ring_pitch, float *__restrict__ density) {
    constexpr int N = P * Q;
    using LQ = zdfft::LineQ<P, E, Q, 1, true>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = LQ::T;
    const int t = threadIdx.x % T, n2 = threadIdx.x / T;
    const int y = blockIdx.x, pl = blockIdx.y;
    const cplx *src = dring + ((long long) pl * N + y) * ring_pitch;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = src[Q * (t + T * e) + n2];
        re[e] = v.x;
        im[e] = v.y;
    }
    LQ::run(re, im, t, 0, n2, lds, twP, twN, twQ);
    float *d0 = density + (2 * (long long) pl * N + y) * N, *d1 = d0 + (long long) N * N;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int xx = (t + T * e) + P * n2;
        d0[xx] = (float) re[e];
        d1[xx] = (float) im[e];
    }
}

namespace zd {

template <int P, int E, int Q>
static int launch_xdens_q_t(const cplx *tw, const void *dring, int ring_pitch, int nplanes, float *density, hipStream_t st) {
    constexpr int N = P * Q, threads = Q * P / E;
    using LQ = zdfft::LineQ<P, E, Q, 1, true>;
    const size_t shmem = sizeof(double) * (size_t) LQ::LDS_DOUBLES;
    if (shmem > 160 * 1024) return 2;
    set_dyn_lds<k_xdens_q<P, E, Q>>(shmem);
    hipLaunchKernelGGL((k_xdens_q<P, E, Q>), dim3(N, nplanes), dim3(threads), shmem, st, tw, tw + P, tw + P + N, (const cplx *) dring, ring_pitch,
                       density);
    ZD_LAUNCH_CHECK();
    return 0;
}

// ---- launchers of the field-store pipeline ----
template <int P, int E, int Q, int NC>
static int launch_zfft_fq_t(const FieldLayout &F, const StoreLayout &S, int ky0, int kyloc0, int nky, const void *Y, const cplx *tw,
                            void *out, hipStream_t st) {
    constexpr int W = NC * FIELD_RB, threads = W * Q * P / E;
    static_assert(threads <= 1024, "workgroup too large");
    const size_t shmem = sizeof(double) * zdfft::LineQ<P, E, Q, W, false>::LDS_DOUBLES;
    if (shmem > 160 * 1024 || nky % FIELD_RB || kyloc0 % FIELD_RB) return 2;
    set_dyn_lds<k_zfft_fq<P, E, Q, NC>>(shmem);
    dim3 grid(S.N / NC, nky / FIELD_RB, F.nfield), block(threads);
    hipLaunchKernelGGL((k_zfft_fq<P, E, Q, NC>), grid, block, shmem, st, F, S, ky0, kyloc0, nky, (const cplx *) Y, tw, tw + P,
                       tw + P + P * Q, (cplx *) out);
    ZD_LAUNCH_CHECK();
    return 0;
}
// tw: exp(2 pi i k/P) | exp(2 pi i k/L) | exp(2 pi i k/Q), consecutive (np2_twiddle_count entries)
int launch_zfft_fields_np2(int L, const FieldLayout &F, const StoreLayout &S, int ky0, int kyloc0, int nky, const void *Y,
                           const void *tw, void *out, hipStream_t st) {
#define ZC(p, q, nc) \
    if (L == (p) * (q)) return launch_zfft_fq_t<p, 16, q, nc>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    ZC(16, 3, 4) ZC(32, 3, 4) ZC(64, 3, 4) ZC(128, 3, 2) ZC(256, 3, 1) ZC(512, 3, 1)
    ZC(16, 9, 4) ZC(32, 9, 2) ZC(64, 9, 1) ZC(128, 9, 1)
    ZC(16, 27, 2) ZC(32, 27, 1) ZC(64, 27, 1)
    // radix 5 (round 3): 5-smooth PPDs such as 1280, 2560, 3200, 3840, 4000, 4800, 5120, 5760, 6400
    ZC(16, 5, 4) ZC(32, 5, 4) ZC(64, 5, 2) ZC(128, 5, 1) ZC(256, 5, 1)
    ZC(16, 15, 4) ZC(32, 15, 2) ZC(64, 15, 1) ZC(128, 15, 1)
    ZC(16, 25, 2) ZC(32, 25, 1) ZC(64, 25, 1)
    ZC(16, 45, 1) ZC(32, 45, 1)
    ZC(16, 75, 1)
    ZC(16, 125, 1)
    // radix 7 (round 4): 224 ... 7168 (Q = 7), 672 ... 5376 (21), 1120 ... 4480 (35), 1568 ... 6272 (49)  (4320 = 32 * 135 and 8640 have z lines
    // of 720, 432, 240, 180, 144: a workgroup of 8 rows x 135 sub-lines would be 1080 threads)
    ZC(16, 7, 4) ZC(32, 7, 4) ZC(64, 7, 4) ZC(128, 7, 2) ZC(256, 7, 1)
    ZC(16, 21, 4) ZC(32, 21, 2) ZC(64, 21, 1)
    ZC(16, 35, 2) ZC(32, 35, 1)
    ZC(16, 49, 2) ZC(32, 49, 1)
#undef ZC
    // 4 * Q: four elements per thread, one thread per sub-line — the short z lines of large stream factors (108 = 4 * 27: PPD = 6912
    // at R = 64, the whole grid on ONE GPU; 500 = 4 * 125: PPD = 4000 at R = 8)
    if (L == 108) return launch_zfft_fq_t<4, 4, 27, 2>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 500) return launch_zfft_fq_t<4, 4, 125, 1>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 300) return launch_zfft_fq_t<4, 4, 75, 1>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 180) return launch_zfft_fq_t<4, 4, 45, 2>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 100) return launch_zfft_fq_t<4, 4, 25, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 60) return launch_zfft_fq_t<4, 4, 15, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    // 8 * Q (round 5): eight elements per thread, one thread per sub-line — the half-length lines of the density-only passes that
    // precede the PLT passes of a PLT + ZD_qdensity = 1 run (zd_capi.cpp plt_dens_split: stream factors R and 2R; PPD = 3456 at R = 8 needs
    // 216 = 8 * 27), and more stream factors to choose from in general
    if (L == 24) return launch_zfft_fq_t<8, 8, 3, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 40) return launch_zfft_fq_t<8, 8, 5, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 56) return launch_zfft_fq_t<8, 8, 7, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 72) return launch_zfft_fq_t<8, 8, 9, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 120) return launch_zfft_fq_t<8, 8, 15, 4>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    if (L == 200) return launch_zfft_fq_t<8, 8, 25, 2>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    // (216 = PPD=6912 ZA at R = 32: ONE column per workgroup — 216 threads = one wave per SIMD, which fits beside the generator's three
    // workgroups of 128 registers on a CU; with two columns, 432 threads, the z FFT only ran when a generator launch drained: Z stage
    // 10.16 -> 9.42 s, profiles/r05_6912_z_stage.txt)
    if (L == 216) return launch_zfft_fq_t<8, 8, 27, 1>(F, S, ky0, kyloc0, nky, Y, (const cplx *) tw, out, st);
    fprintf(stderr, "zeldovich_hip: no z transform of length %d (16*2^k * {3, 9, 27, 5, 15, 25, 45, 75, 125, 7, 21, 35, 49}, or one of the short 4 * Q / 8 * Q lengths)\n", L);
    return 2;
}
// columns per z-FFT workgroup (the generator prunes by it) = the NC of the table above
int zfft_fields_np2_columns(int L) {
    switch (L) {
        case 48: case 96: case 192: case 144: case 80: case 160: case 240: case 100: case 60: case 112: case 224: case 448: case 336: return 4;
        case 24: case 40: case 56: case 72: case 120: return 4;  // 8 * Q
        case 384: case 288: case 432: case 108: case 320: case 480: case 400: case 180: case 896: case 672: case 560: case 784: return 2;
        case 200: return 2;  // 8 * Q (216: one column, see launch_zfft_fields_np2)
        default: return 1;
    }
}

template <int P, int E, int Q, int W>
static int launch_yfft_fq_t(const FieldLayout &F, const StoreLayout &S, const cplx *tw, const void *store, int plane0, int nplanes,
                            int ring_pitch, void *ring, int dens, hipStream_t st) {
    constexpr int threads = W * Q * P / E, N = P * Q;
    static_assert(threads <= 1024, "workgroup too large");
    const size_t shmem = sizeof(double) * zdfft::LineQ<P, E, Q, W, false>::LDS_DOUBLES;
    if (shmem > 160 * 1024) return 2;
    set_dyn_lds<k_yfft_fq<P, E, Q, W>>(shmem);
    dim3 grid((dens ? 1 : 3) * (N / W), 1, nplanes), block(threads);
    hipLaunchKernelGGL((k_yfft_fq<P, E, Q, W>), grid, block, shmem, st, F, S, tw, tw + P, tw + P + N, (const cplx *) store, plane0,
                       ring_pitch, (cplx *) ring, dens);
    ZD_LAUNCH_CHECK();
    return 0;
}
template <int P, int E, int Q, bool PLT>
static int launch_xfft_q_t(const EpiConst &ec, const cplx *tw, const void *ring, int ring_pitch, int nplanes, int z_first, int z_step,
                           void *records, Reduce *red, hipStream_t st) {
    constexpr int N = P * Q, threads = Q * P / E;
    // The one-row form (k_xfft_seq1_q) also where three lines fit a workgroup?  Measured per size, x stage in ms, three-line / one-row:
    // 3456 (128 * 27) 883 / 634, 5120 (1024 * 5) 2978 / 2460 — taken; 2560 322 / 355, 2880 513 / 565, 3072 474 / 512, 3200 673 / 735,
    // 4000 1371 / 1529 — left (it depends on how many workgroups of each form a CU holds: registers, LDS and thread count differ by
    // (P, Q)).  -DZD_SEQ1_MIN=n (make variant) switches every size >= n for such measurements.
#ifdef ZD_SEQ1_MIN
    constexpr bool prefer_seq1 = !PLT && N >= ZD_SEQ1_MIN;
#else
    constexpr bool prefer_seq1 = !PLT && (N == 3456 || N == 5120);
#endif
    if constexpr (3 * threads <= 1024 && !prefer_seq1) {  // three lines per workgroup
        using LQ3 = zdfft::LineQ<P, E, Q, 3, true>;
        constexpr size_t stash = PLT ? 4 * N : 2 * N;  // doubles: the one (ZA) or two (PLT) lines the records read from LDS
        constexpr size_t dbl = LQ3::LDS_DOUBLES > stash ? LQ3::LDS_DOUBLES : stash;
        const size_t shmem3 = sizeof(double) * (dbl > (size_t) 18 * threads ? dbl : (size_t) 18 * threads);
        if (shmem3 <= 160 * 1024) {
            set_dyn_lds<k_xfft_q3<P, E, Q, PLT>>(shmem3);
            hipLaunchKernelGGL((k_xfft_q3<P, E, Q, PLT>), dim3(N, nplanes), dim3(3 * threads), shmem3, st, ec, tw, tw + P, tw + P + N,
                               (const cplx *) ring, ring_pitch, z_first, z_step, (char *) records, red);
            ZD_LAUNCH_CHECK();
            return 0;
        }
    }
    using LQ = zdfft::LineQ<P, E, Q, 1, true>;
    const size_t need = sizeof(double) * (size_t) LQ::LDS_DOUBLES, red_b = sizeof(double) * 6 * threads;
    if constexpr (!PLT) {  // ZA ring: one launch, Im of array 2 parked behind the transform's area
        const size_t need1 = need + sizeof(double) * N, shmem1 = need1 > red_b ? need1 : red_b;
        if (shmem1 <= 160 * 1024) {
            set_dyn_lds<k_xfft_seq1_q<P, E, Q>>(shmem1);
            hipLaunchKernelGGL((k_xfft_seq1_q<P, E, Q>), dim3(N, nplanes), dim3(threads), shmem1, st, ec, tw, tw + P, tw + P + N,
                               (const cplx *) ring, ring_pitch, z_first, z_step, (char *) records, red);
            ZD_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t shmem = need > red_b ? need : red_b;
    if (shmem > 160 * 1024) return 2;
    set_dyn_lds<k_xfft_seq_q<P, E, Q, PLT>>(shmem);
    for (int emit = 0; emit < 2; emit++) {
        dim3 grid(N, nplanes, (emit != 0) == PLT ? 1 : 2), block(threads);
        hipLaunchKernelGGL((k_xfft_seq_q<P, E, Q, PLT>), grid, block, shmem, st, ec, tw, tw + P, tw + P + N, (cplx *) ring, emit, ring_pitch,
                           z_first, z_step, (char *) records, red);
        ZD_LAUNCH_CHECK();
    }
    return 0;
}
// the PPDs with a y / x transform: N = P*Q
#define NP2_SIZES(X)                                                                                                   \
    X(32, 3, 16) X(64, 3, 16) X(128, 3, 16) X(256, 3, 8) X(512, 3, 8) X(1024, 3, 4) X(32, 9, 16) X(64, 9, 16) X(128, 9, 8) \
    X(256, 9, 4) X(512, 9, 2) X(32, 27, 8) X(64, 27, 8) X(128, 27, 4) X(256, 27, 2)                                       \
    /* radix 5 (round 3): 160 ... 5120, 480 ... 7680, 800 ... 6400, 1440 ... 5760, 2400, 4800, 4000, 8000 */               \
    X(32, 5, 16) X(64, 5, 16) X(128, 5, 16) X(256, 5, 8) X(512, 5, 4) X(1024, 5, 2)                                        \
    X(32, 15, 16) X(64, 15, 16) X(128, 15, 8) X(256, 15, 4) X(512, 15, 2)                                                  \
    X(32, 25, 16) X(64, 25, 8) X(128, 25, 4) X(256, 25, 2)                                                                 \
    X(32, 45, 8) X(64, 45, 4) X(128, 45, 2)                                                                                \
    X(32, 75, 4) X(64, 75, 2)                                                                                              \
    X(32, 125, 4) X(64, 125, 2)                                                                                            \
    /* radix 7 (round 4): 224 ... 7168, 672 ... 5376, 1120 ... 4480, 1568 ... 6272; 4320, 8640 = 2^a * 135 */                   \
    X(32, 7, 16) X(64, 7, 16) X(128, 7, 16) X(256, 7, 8) X(512, 7, 4) X(1024, 7, 2)                                        \
    X(32, 21, 16) X(64, 21, 8) X(128, 21, 4) X(256, 21, 2)                                                                 \
    X(32, 35, 8) X(64, 35, 4) X(128, 35, 2)                                                                                \
    X(32, 49, 8) X(64, 49, 4) X(128, 49, 2)                                                                                \
    X(32, 135, 2) X(64, 135, 1)
int launch_yfft_fields_np2(const FieldLayout &F, const StoreLayout &S, const void *tw, const void *store, int plane0, int nplanes,
                           int ring_pitch, void *ring, int dens, hipStream_t st) {
#define YC(p, q, w) \
    if (S.N == (p) * (q)) return launch_yfft_fq_t<p, 16, q, w>(F, S, (const cplx *) tw, store, plane0, nplanes, ring_pitch, ring, dens, st);
    NP2_SIZES(YC)
#undef YC
    fprintf(stderr, "zeldovich_hip: PPD %d is not among the supported 2^a 3^b sizes\n", S.N);
    return 2;
}
int launch_xfft_np2(int N, const EpiConst &ec, const void *tw, const void *ring, int ring_pitch, int nplanes, int z_first, int z_step,
                    void *records, Reduce *red, hipStream_t st) {
#define XC(p, q, w)                                                                                                          \
    if (N == (p) * (q))                                                                                                      \
        return ec.pack == PACK_PLT3                                                                                          \
                   ? launch_xfft_q_t<p, 16, q, true>(ec, (const cplx *) tw, ring, ring_pitch, nplanes, z_first, z_step, records, red, st) \
                   : launch_xfft_q_t<p, 16, q, false>(ec, (const cplx *) tw, ring, ring_pitch, nplanes, z_first, z_step, records, red, st);
    NP2_SIZES(XC)
#undef XC
    fprintf(stderr, "zeldovich_hip: PPD %d is not among the supported 2^a 3^b sizes\n", N);
    return 2;
}
// density planes of `nplanes` store planes from the density ring (k_xdens_q)
int launch_xdens_np2(int N, const void *tw, const void *dring, int ring_pitch, int nplanes, float *density, hipStream_t st) {
#define DC(p, q, w) \
    if (N == (p) * (q)) return launch_xdens_q_t<p, 16, q>((const cplx *) tw, dring, ring_pitch, nplanes, density, st);
    NP2_SIZES(DC)
#undef DC
    fprintf(stderr, "zeldovich_hip: no density x pass for PPD = %d on the composite kernels (unsupported size, or its line does not fit the LDS)\n", N);
    return 2;
}
bool np2_supported_ppd(int N) {
#define SC(p, q, w) \
    if (N == (p) * (q)) return true;
    NP2_SIZES(SC)
#undef SC
    return false;
}
bool np2_supported_zlen(int L) {
    int P, Q;
    if (L == 108 || L == 500 || L == 300 || L == 180 || L == 100 || L == 60) return true;  // 4 * Q (launch_zfft_fields_np2)
    if (L == 24 || L == 40 || L == 56 || L == 72 || L == 120 || L == 200 || L == 216) return true;  // 8 * Q
    if (!np2_split(L, &P, &Q) || P < 16) return false;
    return (Q == 3 && P <= 512) || (Q == 9 && P <= 128) || (Q == 27 && P <= 64) || (Q == 5 && P <= 256) || (Q == 15 && P <= 128)
           || (Q == 25 && P <= 64) || (Q == 45 && P <= 32) || (Q == 75 && P <= 16) || (Q == 125 && P <= 16)
           || (Q == 7 && P <= 256) || (Q == 21 && P <= 64) || (Q == 35 && P <= 32) || (Q == 49 && P <= 32);
}

#ifdef ZD_TESTING
template <int P, int E, int Q, int W>
static int launch_test_fftq_t(int kind, const void *twP, const void *twN, const void *twQ, const void *in, void *out, long long lines,
                              hipStream_t st) {
    constexpr int threads = W * Q * P / E;
    static_assert(threads <= 1024, "workgroup too large");
    if (lines % W) return 3;
    if (kind == 1) {
        const size_t shmem = sizeof(double) * zdfft::LineQ<P, E, Q, W, false>::LDS_DOUBLES;
        set_dyn_lds<k_test_fftq_cols<P, E, Q, W>>(shmem);
        hipLaunchKernelGGL((k_test_fftq_cols<P, E, Q, W>), dim3((unsigned) (lines / W)), dim3(threads), shmem, st, (const cplx *) twP,
                           (const cplx *) twN, (const cplx *) twQ, (const cplx *) in, (cplx *) out, lines);
    } else {
        const size_t shmem = sizeof(double) * zdfft::LineQ<P, E, Q, W, true>::LDS_DOUBLES;
        set_dyn_lds<k_test_fftq_lines<P, E, Q, W>>(shmem);
        hipLaunchKernelGGL((k_test_fftq_lines<P, E, Q, W>), dim3((unsigned) (lines / W)), dim3(threads), shmem, st, (const cplx *) twP,
                           (const cplx *) twN, (const cplx *) twQ, (const cplx *) in, (cplx *) out, lines);
    }
    ZD_LAUNCH_CHECK();
    return 0;
}
#endif  // ZD_TESTING

// n = P*Q -> (P, Q): Q = the whole odd part of n (3^a 5^b 7^c, one of the products below), P the power of two
bool np2_split(int n, int *P, int *Q) {
    int q = 1, p = n;
    while (p % 3 == 0) {
        p /= 3;
        q *= 3;
    }
    while (p % 5 == 0) {
        p /= 5;
        q *= 5;
    }
    while (p % 7 == 0) {
        p /= 7;
        q *= 7;
    }
    const bool known = q == 3 || q == 9 || q == 27 || q == 5 || q == 15 || q == 25 || q == 45 || q == 75 || q == 125 || q == 135 || q == 7
                       || q == 21 || q == 35 || q == 49;
    if (!known || p < 4 || (p & (p - 1)) != 0) return false;
    *P = p;
    *Q = q;
    return true;
}
#ifdef ZD_TESTING
int test_fftq_tile_width(int n) {
    int P, Q;
    if (!np2_split(n, &P, &Q)) return 0;
    return (n / 16) * 4 <= 1024 ? 4 : ((n / 16) * 2 <= 1024 ? 2 : 1);
}

int launch_test_fftq(int n, int kind, const void *twP, const void *twN, const void *twQ, const void *in, void *out, long long lines,
                     hipStream_t st) {
#define TC(p, e, q, w) \
    if (n == (p) * (q)) return launch_test_fftq_t<p, e, q, w>(kind, twP, twN, twQ, in, out, lines, st);
    TC(8, 8, 3, 4) TC(8, 8, 9, 4) TC(8, 8, 27, 4)
    TC(16, 16, 3, 4) TC(16, 16, 9, 4) TC(16, 16, 27, 4)
    TC(32, 16, 3, 4) TC(32, 16, 9, 4) TC(32, 16, 27, 4)
    TC(64, 16, 3, 4) TC(64, 16, 9, 4) TC(64, 16, 27, 4)
    TC(128, 16, 3, 4) TC(128, 16, 9, 4) TC(128, 16, 27, 4)
    TC(256, 16, 3, 4) TC(256, 16, 9, 4) TC(256, 16, 27, 2)
    TC(512, 16, 3, 4) TC(512, 16, 9, 2)
    TC(16, 16, 5, 4) TC(32, 16, 5, 4) TC(256, 16, 5, 4) TC(1024, 16, 5, 2)
    TC(16, 16, 15, 4) TC(64, 16, 15, 4) TC(16, 16, 25, 4) TC(128, 16, 25, 4) TC(16, 16, 45, 4) TC(32, 16, 75, 4) TC(16, 16, 125, 4) TC(32, 16, 125, 4)
    TC(8, 8, 5, 4) TC(8, 8, 25, 4) TC(8, 8, 125, 4)
    TC(8, 8, 7, 4) TC(16, 16, 7, 4) TC(64, 16, 7, 4) TC(256, 16, 7, 4) TC(1024, 16, 7, 2) TC(16, 16, 21, 4) TC(128, 16, 21, 4) TC(16, 16, 35, 4)
    TC(64, 16, 35, 4) TC(16, 16, 49, 4) TC(128, 16, 49, 2) TC(16, 16, 135, 4) TC(32, 16, 135, 2) TC(64, 16, 135, 1)
    TC(1024, 16, 3, 4)
#undef TC
    fprintf(stderr, "zeldovich_hip: no composite FFT for length %d\n", n);
    return 2;
}
#endif  // ZD_TESTING

}  // namespace zd
