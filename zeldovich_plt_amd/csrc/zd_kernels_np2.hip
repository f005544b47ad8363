// zd_kernels_np2.hip — kernels for PPD = 2^a 3^b (not a power of two): the composite-length line transform of zd_fft_q.h
// in the roles of k_zfft_f / k_yfft_f / k_xfft_seq (field store, ZA, one rank).  A separate translation unit so that the
// power-of-two kernels (zd_kernels.hip) and their build time are untouched.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "zd_device.h"
#include "zd_fft_q.h"
#include "zd_launch.h"

using namespace zd;
using zdfft::cplx;

#define ZD_LAUNCH_CHECK()                                                                       \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "zeldovich_hip: launch failed at %s:%d: %s\n", __FILE__, __LINE__,  \
                    hipGetErrorString(e__));                                                    \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

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

namespace zd {

template <int P, int E, int Q, int W>
static int launch_test_fftq_t(int kind, const void *twP, const void *twN, const void *twQ, const void *in, void *out, long long lines,
                              hipStream_t st) {
    constexpr int threads = W * Q * P / E;
    static_assert(threads <= 1024, "workgroup too large");
    if (lines % W) return 3;
    if (kind == 1) {
        const size_t shmem = sizeof(double) * zdfft::LineQ<P, E, Q, W, false>::LDS_DOUBLES;
        hipFuncSetAttribute((const void *) k_test_fftq_cols<P, E, Q, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) shmem);
        hipLaunchKernelGGL((k_test_fftq_cols<P, E, Q, W>), dim3((unsigned) (lines / W)), dim3(threads), shmem, st, (const cplx *) twP,
                           (const cplx *) twN, (const cplx *) twQ, (const cplx *) in, (cplx *) out, lines);
    } else {
        const size_t shmem = sizeof(double) * zdfft::LineQ<P, E, Q, W, true>::LDS_DOUBLES;
        hipFuncSetAttribute((const void *) k_test_fftq_lines<P, E, Q, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) shmem);
        hipLaunchKernelGGL((k_test_fftq_lines<P, E, Q, W>), dim3((unsigned) (lines / W)), dim3(threads), shmem, st, (const cplx *) twP,
                           (const cplx *) twN, (const cplx *) twQ, (const cplx *) in, (cplx *) out, lines);
    }
    ZD_LAUNCH_CHECK();
    return 0;
}

// n = P*Q -> (P, Q): Q = the whole power of three in n (3, 9 or 27), P the power of two
bool np2_split(int n, int *P, int *Q) {
    int q = 1, p = n;
    while (p % 3 == 0) {
        p /= 3;
        q *= 3;
    }
    if (q == 1 || q > 27 || p < 8 || (p & (p - 1)) != 0) return false;
    *P = p;
    *Q = q;
    return true;
}
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
    TC(1024, 16, 3, 4)
#undef TC
    fprintf(stderr, "zeldovich_hip: no composite FFT for length %d\n", n);
    return 2;
}

}  // namespace zd
