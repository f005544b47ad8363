// zd_fft_q.h — line lengths N = P * Q with P a power of two and Q = 3, 9, 27 (PPD = 2^a 3^b: e.g. 6912 = 256 * 27), on
// top of the register-resident power-of-two engine of zd_fft.h.
//
// The reference plans any length with FFTW (src/zeldovich.cpp:61-66).  Here a line is split into its Q decimated
// sub-sequences x[Q n1 + n2] (n2 < Q), each of length P:
//     X[k1 + P k2] = sum_{n2 < Q}  ( W_N^{n2 k1}  F_{n2}[k1] )  W_Q^{n2 k2},      F_{n2} = DFT_P of sub-sequence n2
// The Q sub-transforms are just Q more "columns" for the existing engine (threads (t, n2) hold 16 elements each, radix-16
// butterflies in registers, exchanges through LDS); the outer Q-point transforms are log3(Q) radix-3 stages on an LDS
// staging buffer (the power-of-two sizes never come here).  Thread (t, w, n2) enters with elements  x[Q (t + T e) + n2]  of line w and leaves with  X[(t + T e) + P n2].
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
        // outer Q-point transforms (Q = 3^S) as S radix-3 stages on an LDS staging buffer, CH elements per round.
        // Position p = d_0 + 3 d_1 + 9 d_2 of a group holds, before stage s, the partial transform whose digits d_j with
        // j > S-1-s are already output digits; stage s sums over digit j = S-1-s:
        //     out(d) = sum_r in(d with d_j = r) w_{3^(s+1)}^{r e_s},   e_s = n2 mod 3^(s+1)
        // with the thread's own digits taken from n2 = d_{S-1} + 3 d_{S-2} + ... (reversed), so that after the last stage
        // the thread holds exactly output k2 = n2 and nothing has to be exchanged again.
        constexpr int S = Q == 3 ? 1 : (Q == 9 ? 2 : 3);
        static_assert(Q == 3 || Q == 9 || Q == 27, "Q = 3, 9, 27");
        int pos = 0;  // the thread's position p
        {
            int r = n2, w3 = Q / 3;
#pragma unroll
            for (int j = 0; j < S; j++) {
                pos += (r % 3) * w3;
                r /= 3;
                w3 /= 3;
            }
        }
        cplx *stage = reinterpret_cast<cplx *>(lds);
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += CH) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CH; c++) stage[(((c * T + t) * W + w) * Q) + n2] = cplx{re[e0 + c], im[e0 + c]};
#pragma unroll
            for (int s = 0; s < S; s++) {
                constexpr int P3[4] = {1, 3, 9, 27};
                const int stride = Q / P3[s + 1];                  // 3^j, j = S-1-s
                const int base   = pos - ((pos / stride) % 3) * stride;
                const int es     = n2 % P3[s + 1];
                const cplx w1 = twQ[(es * stride) % Q], w2 = twQ[(2 * es * stride) % Q];
                __syncthreads();
                double vr[CH], vi[CH];
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const cplx *g = stage + ((c * T + t) * W + w) * Q + base;
                    const cplx a = g[0], b = g[stride], d = g[2 * stride];
                    vr[c] = a.x + (b.x * w1.x - b.y * w1.y) + (d.x * w2.x - d.y * w2.y);
                    vi[c] = a.y + (b.x * w1.y + b.y * w1.x) + (d.x * w2.y + d.y * w2.x);
                }
                if (s + 1 < S) {
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < CH; c++) stage[((c * T + t) * W + w) * Q + pos] = cplx{vr[c], vi[c]};
                } else {
#pragma unroll
                    for (int c = 0; c < CH; c++) {
                        re[e0 + c] = vr[c];
                        im[e0 + c] = vi[c];
                    }
                }
            }
        }
        __syncthreads();
    }
};
#endif

}  // namespace zdfft
