// zd_fft_q.h — line lengths N = P * Q with P a power of two and Q = 3^a 5^b 7^c (3, 9, 27, 5, 25, 125, 15, 45, 75, 135, 7, 21, 35, 49;
// PPD = 6912 = 256 * 27, 4000 = 32 * 125, 7168 = 1024 * 7, ...), on
// top of the register-resident power-of-two engine of zd_fft.h.
//
// The reference plans any length with FFTW (src/zeldovich.cpp:61-66).  Here a line is split into its Q decimated
// sub-sequences x[Q n1 + n2] (n2 < Q), each of length P:
//     X[k1 + P k2] = sum_{n2 < Q}  ( W_N^{n2 k1}  F_{n2}[k1] )  W_Q^{n2 k2},      F_{n2} = DFT_P of sub-sequence n2
// The Q sub-transforms are just Q more "columns" for the existing engine (threads (t, n2) hold 16 elements each, radix-16
// butterflies in registers, exchanges through LDS); the outer Q-point transforms are a + b + c radix-3 / radix-5 / radix-7 stages on an LDS
// staging buffer (the power-of-two sizes never come here).  Thread (t, w, n2) enters with elements  x[Q (t + T e) + n2]  of line w and leaves with  X[(t + T e) + P n2].
#pragma once
#include <type_traits>

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

// Q = 3^a 5^b 7^c: digit j of an index below Q has radix 3 for j < a, 5 for the next b digits and 7 above
template <int Q>
struct QFactors {
    static constexpr int count(int q, int r) { return q % r == 0 ? 1 + count(q / r, r) : 0; }
    static constexpr int A = count(Q, 3), B = count(Q, 5), C = count(Q, 7), S = A + B + C;
    static constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }
    static_assert(ipow(3, A) * ipow(5, B) * ipow(7, C) == Q && S >= 1 && S <= 4, "Q = 3^a 5^b 7^c with 1 <= a + b + c <= 4");
    static constexpr int radix(int j) { return j < A ? 3 : (j < A + B ? 5 : 7); }
    static constexpr int stride(int j) { return j == 0 ? 1 : stride(j - 1) * radix(j - 1); }  // S_j
};

#if defined(__HIPCC__)
// the stages of the outer transform over the digits J, J - 1, ..., 0 (see LineQ::run)
template <class QF, int J, int E, int CH, int T, int W, int Q>
__device__ __forceinline__ void outer_stages(int k, int t, int w, int e0, cplx *stage, const cplx *__restrict__ twQ, double (&re)[E],
                                             double (&im)[E]) {
    constexpr int R = QF::radix(J), Sj = QF::stride(J), Mj = Q / Sj, Mn = Mj / R;
    const int low = k / Mj, kap = k % Mj, kapn = k % Mn;  // (Mn = 1 at the first stage: kapn = 0)
    const int rd = low + Sj * R * kapn;                  // + Sj * r: the R inputs of this stage
    cplx wq[R];
#pragma unroll
    for (int r = 1; r < R; r++) wq[r] = twQ[(r * kap * Sj) % Q];
    __syncthreads();
    double vr[CH], vi[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const cplx *g = stage + ((c * T + t) * W + w) * Q + rd;
        const cplx a0 = g[0];
        double ar = a0.x, ai = a0.y;
#pragma unroll
        for (int r = 1; r < R; r++) {
            const cplx b = g[r * Sj];
            ar += b.x * wq[r].x - b.y * wq[r].y;
            ai += b.x * wq[r].y + b.y * wq[r].x;
        }
        vr[c] = ar;
        vi[c] = ai;
    }
    if constexpr (J > 0) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; c++) stage[((c * T + t) * W + w) * Q + low + Sj * kap] = cplx{vr[c], vi[c]};
        outer_stages<QF, J - 1, E, CH, T, W, Q>(k, t, w, e0, stage, twQ, re, im);
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            re[e0 + c] = vr[c];
            im[e0 + c] = vi[c];
        }
    }
}

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
        // Outer Q-point transforms, Q = 3^a 5^b 7^c, as a + b + c stages of radix 3 / 5 / 7 on an LDS staging buffer, CH elements per round.
        // Input index n = d_0 + r_0 (d_1 + r_1 (d_2 + ...)) (digit j has radix r_j and weight S_j = r_0 ... r_{j-1}), so
        //     X[k] = sum_{d_0} W_Q^{d_0 k}  sum_{d_1} W_{Q/S_1}^{d_1 k}  ...  sum_{d_{S-1}} W_{r_{S-1}}^{d_{S-1} k}
        // and stage s sums over the digit j = S-1-s (the innermost sum first) with the twiddle W_{M_j}^{d_j k}, M_j = Q / S_j,
        // which depends on k mod M_j only.  After it the partial result is a function of (d_0 .. d_{j-1}, k mod M_j) and is kept
        // at position (d_0 .. d_{j-1} as a number below S_j) + S_j (k mod M_j).  The thread with n2 = k computes, at every stage,
        // the entry whose low digits are the number k / M_j: (k / M_j, k mod M_j) is a one-to-one image of k, and at the last
        // stage (j = 0, M_0 = Q) the thread is left with exactly X[k] — nothing has to be exchanged again.
        using QF = QFactors<Q>;
        constexpr int S = QF::S;
        const int k = n2;
        cplx *stage = reinterpret_cast<cplx *>(lds);
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += CH) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CH; c++) stage[(((c * T + t) * W + w) * Q) + n2] = cplx{re[e0 + c], im[e0 + c]};
            outer_stages<QF, S - 1, E, CH, T, W, Q>(k, t, w, e0, stage, twQ, re, im);
        }
        __syncthreads();
    }
};
#endif

}  // namespace zdfft
