// zd_fft_q.h — line lengths N = P * Q with P a power of two and Q = 3, 9, 27 (PPD = 2^a 3^b: e.g. 6912 = 256 * 27), on
// top of the register-resident power-of-two engine of zd_fft.h.
//
// The reference plans any length with FFTW (src/zeldovich.cpp:61-66).  Here a line is split into its Q decimated
// sub-sequences x[Q n1 + n2] (n2 < Q), each of length P:
//     X[k1 + P k2] = sum_{n2 < Q}  ( W_N^{n2 k1}  F_{n2}[k1] )  W_Q^{n2 k2},      F_{n2} = DFT_P of sub-sequence n2
// The Q sub-transforms are just Q more "columns" for the existing engine (threads (t, n2) hold 16 elements each, radix-16
// butterflies in registers, exchanges through LDS); the outer Q-point transforms are done as Q complex multiply-adds per
// output straight from an LDS staging buffer (O(Q) per output: Q <= 27, a compatibility path — the power-of-two sizes never
// come here).  Thread (t, w, n2) enters with elements  x[Q (t + T e) + n2]  of line w and leaves with  X[(t + T e) + P n2].
#pragma once
#include "zd_fft.h"

namespace zdfft {

template <bool C, class A, class B>
struct pick {
    using type = A;
};
template <class A, class B>
struct pick<false, A, B> {
    using type = B;
};

#if defined(__HIPCC__)
// W lines per workgroup, column of the sub-engine = w + W * n2 (w fastest: global accesses coalesce along w);
// LINE = true: LineInner layout (contiguous lines, x pass), column = line index.
//   twP: exp(2 pi i k / P), k < P;  twN: exp(2 pi i k / (P Q)), k < P Q;  twQ: exp(2 pi i k / Q), k < Q
// `lds` must hold max(sub-engine policy SIZE, CH * T * W * Q * 2) doubles (fftq_lds_doubles below).
template <int P, int E, int Q, int W, bool LINE>
struct LineQ {
    using PL  = Plan<P, E>;
    using LDS = typename pick<LINE, LineInner<P, W * Q>, ColsInner<P, W * Q>>::type;
    static constexpr int T  = PL::T;
    static constexpr int CH = E >= 4 ? 4 : E;  // elements combined per staging round
    static constexpr int STAGE = CH * T * W * Q * 2;
    static constexpr int LDS_DOUBLES = LDS::SIZE > STAGE ? LDS::SIZE : STAGE;

    static __device__ __forceinline__ void run(double (&re)[E], double (&im)[E], int t, int w, int n2, double *lds,
                                               const cplx *__restrict__ twP, const cplx *__restrict__ twN,
                                               const cplx *__restrict__ twQ) {
        fft_line<PL, LDS>(re, im, t, w + W * n2, lds, twP);
        // twiddle W_N^{n2 k1}
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int k1 = t + T * e;
            const cplx tw = twN[(n2 * k1) % (P * Q)];
            const double a = re[e] * tw.x - im[e] * tw.y, b = re[e] * tw.y + im[e] * tw.x;
            re[e] = a;
            im[e] = b;
        }
        // outer Q-point transforms, CH elements per round: stage[((e - e0) * T + t) * W + w][m] complex
        cplx *stage = reinterpret_cast<cplx *>(lds);
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += CH) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CH; c++) stage[(((c * T + t) * W + w) * Q) + n2] = cplx{re[e0 + c], im[e0 + c]};
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const cplx *g = stage + ((c * T + t) * W + w) * Q;
                double ar = 0.0, ai = 0.0;
                int idx = 0;  // (m * k2) mod Q with k2 = n2
#pragma unroll 3
                for (int m = 0; m < Q; m++) {
                    const cplx v = g[m], wq = twQ[idx];
                    ar += v.x * wq.x - v.y * wq.y;
                    ai += v.x * wq.y + v.y * wq.x;
                    idx += n2;
                    idx = idx >= Q ? idx - Q : idx;
                }
                re[e0 + c] = ar;
                im[e0 + c] = ai;
            }
        }
        __syncthreads();
    }
};
#endif

}  // namespace zdfft
