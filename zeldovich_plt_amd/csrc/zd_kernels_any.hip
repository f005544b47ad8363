// zd_kernels_any.hip — ANY even PPD (the reference plans every length with FFTW, src/zeldovich.cpp:61-66; its only
// conditions are an even ppd divisible by NumBlock, src/parameters.cpp:123-126, src/block_array.cpp:38-40).
//
// Powers of two run on the register engine (zd_fft.h), 2^a 3^b on the composite kernels (zd_kernels_np2.hip).  Everything
// else — 1000, 1250, 1440, 5000, 2 x a prime ... — comes here: a length-n transform as a CONVOLUTION of length M = 2^m >=
// 2n - 1 (Bluestein), which the power-of-two engine does in registers:
//     X_k = sum_n x_n e^{+2 pi i nk/n} = c_k sum_n (x_n c_n) conj(c_{k-n}),      c_m = e^{+i pi m^2 / n}
//     a_i = x_i c_i (i < n, else 0);   A = F_M a;   C_i = A_i FB_i  (FB = F_M of b_m = conj(c_m), |m| < n, wrapped);
//     X_k = c_k (F_M^{-1} C)_k,  k < n
// i.e. two engine transforms and three pointwise products per line; the forward transform is conj o inverse o conj.  A
// thread holds the same indices t + T e before and after an engine transform, so the pointwise steps need no exchange.
// It is a compatibility path (about 6x the arithmetic of a native size and lines padded to M): reference arrays
// (include/block_array.h:26-35: 1 / 2 / 4 complex arrays with Hermitian twins), one rank, every option (ZD_f_NL: the phi
// round's forward transform of a real field is the conjugate of its inverse transform, k_any_phi_nl / k_any_phik).
// The pipeline is cut into simple pieces:
//     generator (k_genf / k_gen, zd_kernels.hip)  ->  Y[job][row][k2][x]
//     k_any_cols   in-place transform of strided lines: Y along k2 (z stage), the store along y (y stage)
//     k_any_scatter  Y -> store rows ky / N - ky with the Hermitian twin rules of k_zfft
//     k_any_lines  in-place transform of contiguous lines (x stage)
//     k_any_emit   WriteParticlesSlab (src/output.cpp:86-203): records, density, reductions
// Store layout: [plane z2][array a][row y][x], row pitch N (+ pad).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zd_device.h"
#include "zd_epi.h"
#include "zd_launch.h"

using namespace zd;
using zdfft::cplx;

namespace {

// one Bluestein transform of the line whose elements this thread holds (indices i = t + T e of the length-M work line):
// on entry re/im = x_i for i < n (anything for i >= n), on exit X_i for i < n
template <class PL, class LDS>
__device__ __forceinline__ void bluestein(double (&re)[PL::E], double (&im)[PL::E], int t, int w, double *lds, const AnyTab &tb) {
    constexpr int E = PL::E, T = PL::T;
#pragma unroll
    for (int e = 0; e < E; e++) {  // a_i = x_i c_i, conjugated for the forward transform
        const int i = t + T * e;
        double ar = 0.0, ai = 0.0;
        if (i < tb.n) {
            const cplx c = tb.chirp[i];
            ar = re[e] * c.x - im[e] * c.y;
            ai = re[e] * c.y + im[e] * c.x;
        }
        re[e] = ar;
        im[e] = -ai;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tb.twM);  // conj(A)
#pragma unroll
    for (int e = 0; e < E; e++) {  // C = A FB
        const cplx f = tb.fb[t + T * e];
        const double ar = re[e], ai = -im[e];
        re[e] = ar * f.x - ai * f.y;
        im[e] = ar * f.y + ai * f.x;
    }
    __syncthreads();
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tb.twM);  // M times the convolution
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = t + T * e;
        if (i < tb.n) {
            const cplx c = tb.chirp[i];
            const double vr = re[e] * tb.invM, vi = im[e] * tb.invM;
            re[e] = vr * c.x - vi * c.y;
            im[e] = vr * c.y + vi * c.x;
        }
    }
}

}  // namespace

// in-place transform of strided lines: line (batch, x): points data[batch * batch_stride + x + i * point_stride], i < n,
// for the columns x < ncols.  zero_point >= 0: that input point counts as zero (the Nyquist row of the y stage,
// zeldovich.cpp:644-650).     grid: (ceil(ncols / W), nbatch)   block: W * M / E
template <int M, int E, int W>
__global__ __launch_bounds__(W *M / E) void k_any_cols(AnyTab tb, cplx *__restrict__ data, long long batch_stride, long long point_stride,
                                                      int ncols, int zero_point) {
    using PL  = zdfft::Plan<M, E>;
    using LDS = zdfft::ColsInner<M, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    const int x = blockIdx.x * W + w;
    const bool on = x < ncols;
    cplx *base = data + (long long) blockIdx.y * batch_stride + (on ? x : 0);
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = t + T * e;
        cplx v = cplx{0.0, 0.0};
        if (on && i < tb.n && i != zero_point) v = base[(long long) i * point_stride];
        re[e] = v.x;
        im[e] = v.y;
    }
    bluestein<PL, LDS>(re, im, t, w, lds, tb);
    if (!on) return;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = t + T * e;
        if (i < tb.n) base[(long long) i * point_stride] = cplx{re[e], im[e]};
    }
}

// in-place transform of contiguous lines: line l at data[l * pitch], l < nlines.   grid: ceil(nlines / W)   block: W * M / E
template <int M, int E, int W>
__global__ __launch_bounds__(W *M / E) void k_any_lines(AnyTab tb, cplx *__restrict__ data, long long pitch, long long nlines) {
    using PL  = zdfft::Plan<M, E>;
    using LDS = zdfft::LineInner<M, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int t = threadIdx.x % T, w = threadIdx.x / T;
    const long long line = (long long) blockIdx.x * W + w;
    const bool on = line < nlines;
    cplx *base = data + (on ? line : 0) * pitch;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = t + T * e;
        cplx v = cplx{0.0, 0.0};
        if (on && i < tb.n) v = base[i];
        re[e] = v.x;
        im[e] = v.y;
    }
    bluestein<PL, LDS>(re, im, t, w, lds, tb);
    if (!on) return;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = t + T * e;
        if (i < tb.n) base[i] = cplx{re[e], im[e]};
    }
}

// z stage, second half: the transformed folded inputs Y[((j * nky + kyl) * L + z2) * N + x] go to the store rows ky
// ("self") and N - ky, column N - x, conjugated ("twin") — the rules of k_zfft (zd_kernels.hip) / LoadPlane's slab and
// slabHer writes (zeldovich.cpp:447-466).     grid: (ceil(N / 256), L, njobs * nky)   block: 256
__global__ __launch_bounds__(256) void k_any_scatter(JobList jobs, AnyLayout A, int ky0, int nky, int L, const cplx *__restrict__ Y,
                                                    cplx *__restrict__ store) {
    const int N = A.N;
    const int x = blockIdx.x * 256 + threadIdx.x, z2 = blockIdx.y;
    const int j = blockIdx.z / nky, kyl = blockIdx.z % nky;
    if (x >= N) return;
    const int ky = ky0 + kyl;
    const int kind = jobs.kind[j], arr = jobs.arr[j];
    const bool twin_only = jobs.twin[j] != 0;
    if (ky == 0 && twin_only) return;  // ky = 0 is its own twin plane: every column written as "self"
    const cplx v = Y[(((long long) j * nky + kyl) * L + z2) * N + x];
    cplx *plane = store + ((long long) z2 * A.narray + arr) * N * A.pitch;
    if (!twin_only) plane[(long long) ky * A.pitch + x] = v;
    if (ky != 0 && (twin_only || kind == JOB_C_BOTH || kind == JOB_DENS)) {
        const double sgr = (kind == JOB_C_BOTH) ? -1.0 : 1.0, sgi = (kind == JOB_C_BOTH) ? 1.0 : -1.0;  // -conj / conj
        plane[(long long) (N - ky) * A.pitch + (x ? N - x : 0)] = cplx{sgr * v.x, sgi * v.y};
    }
}

// WriteParticlesSlab (src/output.cpp:86-203) from the transformed arrays of planes [plane0, plane0 + nplanes):
//   dens = Re a0; pos = (Im a0, Re a1, Im a1); vel = PLT ? (Im a2, Re a3, Im a3) : pos * vnorm
//   grid: (ceil(N / 256), N, nplanes)   block: 256
template <int NA>
__global__ __launch_bounds__(256) void k_any_emit(AnyLayout A, EpiConst ec, const cplx *__restrict__ store, int plane0, int z_first, int z_step,
                                                 char *__restrict__ records, float *__restrict__ density, Reduce *__restrict__ red) {
    __shared__ double scr[7 * 4];
    const int N = A.N;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pl = plane0 + blockIdx.z;
    const int z = z_first + z_step * (int) blockIdx.z;
    double ssq = 0.0;
    MaxAbs mx;
    if (x < N) {
        const cplx *row = store + (((long long) pl * A.narray) * N + y) * A.pitch + x;
        const long long astride = (long long) N * A.pitch;
        const cplx a0 = row[0];
        const long long pidx = (long long) blockIdx.z * N * N + (long long) y * N + x;
        ssq = a0.x * a0.x;
        if (density) density[pidx] = (float) a0.x;
        if constexpr (NA >= 2) {
            const cplx a1 = row[astride];
            const double pos[3] = {a0.y, a1.x, a1.y};
            double vel[3] = {pos[0] * ec.vnorm, pos[1] * ec.vnorm, pos[2] * ec.vnorm};
            if constexpr (NA == 4) {
                const cplx a2 = row[2 * astride], a3 = row[3 * astride];
                vel[0] = a2.y * ec.vnorm;
                vel[1] = a3.x * ec.vnorm;
                vel[2] = a3.y * ec.vnorm;
            }
            max_track(mx, pos, ((unsigned long long) z * N + (unsigned long long) y) * N + (unsigned long long) x);
            if (records) emit_record(records, pidx, ec, z, y, x, pos, vel);
        }
    }
    xfft_reduce<256, NA>(scr, red, ssq, mx);
}

// ZD_f_NL, phi round (zeldovich.cpp:699-790): phi(x) = Re of the inverse transform -> (phi + f_NL phi^2) / N^3, real
//   grid: (ceil(N / 256), N, N)   block: 256        store [z][y][x], one array
__global__ __launch_bounds__(256) void k_any_phi_nl(AnyLayout A, double f_NL, double inv_ppd3, cplx *__restrict__ store) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= A.N) return;
    cplx *p = store + ((long long) blockIdx.z * A.N + blockIdx.y) * A.pitch + x;
    const double phi = p->x;
    *p = cplx{(phi + f_NL * phi * phi) * inv_ppd3, 0.0};
}
// the forward 3-D transform of a real field is the conjugate of its inverse transform: PhiK[ky][kz][x] = conj store[kz][ky][x]
// for the half-space rows ky < N/2     grid: (ceil(N / 256), N, N / 2)   block: 256
__global__ __launch_bounds__(256) void k_any_phik(AnyLayout A, const cplx *__restrict__ store, cplx *__restrict__ phik) {
    const int x = blockIdx.x * 256 + threadIdx.x, kz = blockIdx.y, ky = blockIdx.z;
    if (x >= A.N) return;
    const cplx v = store[((long long) kz * A.N + ky) * A.pitch + x];
    phik[((long long) ky * A.N + kz) * A.N + x] = cplx{v.x, -v.y};
}

namespace zd {

int launch_any_phi_nl(const AnyLayout &A, double f_NL, void *store, hipStream_t st) {
    const double inv = 1. / A.N / A.N / A.N;
    hipLaunchKernelGGL(k_any_phi_nl, dim3((A.N + 255) / 256, A.N, A.N), dim3(256), 0, st, A, f_NL, inv, (cplx *) store);
    ZD_LAUNCH_CHECK();
    return 0;
}
int launch_any_phik(const AnyLayout &A, const void *store, void *phik, hipStream_t st) {
    hipLaunchKernelGGL(k_any_phik, dim3((A.N + 255) / 256, A.N, A.N / 2), dim3(256), 0, st, A, (const cplx *) store, (cplx *) phik);
    ZD_LAUNCH_CHECK();
    return 0;
}

int any_engine_size(int n) {  // M = 2^m >= 2n - 1, at least 64
    int M = 64;
    while (M < 2 * n - 1) M *= 2;
    return M;
}

template <int M, int W>
static int launch_any_cols_t(const AnyTab &tb, void *data, long long batch_stride, long long point_stride, int ncols, int nbatch,
                             int zero_point, hipStream_t st) {
    constexpr int E = 16, threads = W * M / E;
    const size_t shmem = sizeof(double) * zdfft::ColsInner<M, W>::SIZE;
    set_dyn_lds<k_any_cols<M, E, W>>(shmem);
    hipLaunchKernelGGL((k_any_cols<M, E, W>), dim3((ncols + W - 1) / W, nbatch), dim3(threads), shmem, st, tb, (cplx *) data, batch_stride,
                       point_stride, ncols, zero_point);
    ZD_LAUNCH_CHECK();
    return 0;
}
template <int M, int W>
static int launch_any_lines_t(const AnyTab &tb, void *data, long long pitch, long long nlines, hipStream_t st) {
    constexpr int E = 16, threads = W * M / E;
    const size_t shmem = sizeof(double) * zdfft::LineInner<M, W>::SIZE;
    set_dyn_lds<k_any_lines<M, E, W>>(shmem);
    hipLaunchKernelGGL((k_any_lines<M, E, W>), dim3((unsigned) ((nlines + W - 1) / W)), dim3(threads), shmem, st, tb, (cplx *) data, pitch,
                       nlines);
    ZD_LAUNCH_CHECK();
    return 0;
}

// engine size, columns per workgroup (strided lines), lines per workgroup (contiguous lines)
#define ANY_SIZES(X) X(64, 16, 4) X(128, 16, 4) X(256, 16, 4) X(512, 16, 4) X(1024, 8, 4) X(2048, 8, 4) X(4096, 4, 4) X(8192, 2, 2) X(16384, 1, 1)

int launch_any_cols(const AnyTab &tb, void *data, long long batch_stride, long long point_stride, int ncols, int nbatch, int zero_point,
                    hipStream_t st) {
#define AC(m, wc, wl) \
    if (tb.M == m) return launch_any_cols_t<m, wc>(tb, data, batch_stride, point_stride, ncols, nbatch, zero_point, st);
    ANY_SIZES(AC)
#undef AC
    fprintf(stderr, "zeldovich_hip: no transform engine of size %d\n", tb.M);
    return 2;
}
int launch_any_lines(const AnyTab &tb, void *data, long long pitch, long long nlines, hipStream_t st) {
#define AL(m, wc, wl) \
    if (tb.M == m) return launch_any_lines_t<m, wl>(tb, data, pitch, nlines, st);
    ANY_SIZES(AL)
#undef AL
    fprintf(stderr, "zeldovich_hip: no transform engine of size %d\n", tb.M);
    return 2;
}
int launch_any_scatter(const JobList &jobs, const AnyLayout &A, int ky0, int nky, int L, const void *Y, void *store, hipStream_t st) {
    hipLaunchKernelGGL(k_any_scatter, dim3((A.N + 255) / 256, L, jobs.n * nky), dim3(256), 0, st, jobs, A, ky0, nky, L, (const cplx *) Y,
                       (cplx *) store);
    ZD_LAUNCH_CHECK();
    return 0;
}
int launch_any_emit(const AnyLayout &A, const EpiConst &ec, const void *store, int plane0, int nplanes, int z_first, int z_step, void *records,
                    float *density, Reduce *red, hipStream_t st) {
    dim3 grid((A.N + 255) / 256, A.N, nplanes), block(256);
    if (A.narray == 1) {
        hipLaunchKernelGGL(k_any_emit<1>, grid, block, 0, st, A, ec, (const cplx *) store, plane0, z_first, z_step, (char *) records, density, red);
        ZD_LAUNCH_CHECK();
    } else if (A.narray == 2) {
        hipLaunchKernelGGL(k_any_emit<2>, grid, block, 0, st, A, ec, (const cplx *) store, plane0, z_first, z_step, (char *) records, density, red);
        ZD_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_any_emit<4>, grid, block, 0, st, A, ec, (const cplx *) store, plane0, z_first, z_step, (char *) records, density, red);
        ZD_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace zd
