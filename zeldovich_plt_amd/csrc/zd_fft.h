// zd_fft.h — register-resident radix-E Stockham FFT for gfx950 (fp64 complex, unnormalised, sign +1).
//
// Replaces the FFTW3 calls of the reference (src/zeldovich.cpp:83-114: plan1d/plan2d, sign +1,
// unnormalised).  MI355X-first design: a length-N line is owned by T = N/E threads; each thread keeps
// E complex doubles in VGPRs for the WHOLE transform (element e of thread t is index t + T*e, both
// on entry and on exit — the Stockham autosort makes the distribution a fixed point), does radix-E
// butterflies in registers, and only the inter-pass transposes go through LDS.  With `SPLIT` the real
// and imaginary planes are exchanged one after the other so a tile needs 8 B (not 16 B) of LDS per
// element: that is what lets a 16K-element tile (e.g. 2048 x 8 columns = 128-byte HBM runs) fit
// beside a second workgroup in the 160 KB LDS of a CU.
//
// Everything here is `__host__ __device__` so that tests/ can run the exact same index arithmetic on
// the CPU (tests/host_emul) — there is no GPU in the build container.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZD_HD __host__ __device__ __forceinline__
#else
#define ZD_HD inline __attribute__((always_inline))
#endif

namespace zdfft {

struct cplx {
    double x, y;
};

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

template <int N_, int E_>
struct Plan {
    static_assert((N_ & (N_ - 1)) == 0 && (E_ & (E_ - 1)) == 0 && N_ >= E_ && E_ >= 2, "pow2 only");
    static constexpr int N     = N_;
    static constexpr int E     = E_;
    static constexpr int T     = N_ / E_;  // threads per line
    static constexpr int LOGN  = ilog2c(N_);
    static constexpr int LOGE  = ilog2c(E_);
    static constexpr int NPASS = (LOGN + LOGE - 1) / LOGE;
    // passes 0..NPASS-2 have radix E; the last one takes what is left
    static constexpr int radix(int p) { return (LOGN - p * LOGE) >= LOGE ? E_ : (1 << (LOGN - p * LOGE)); }
    static constexpr int ns(int p) { return 1 << (p * LOGE); }  // product of the earlier radices
};

// ---- twiddles of the in-register butterflies: W16^k = exp(+i pi k / 8), k = 0..7 --------------
#define ZD_C1 0.92387953251128673848  // cos(pi/8)
#define ZD_S1 0.38268343236508978178  // sin(pi/8)
#define ZD_R2 0.70710678118654752440  // sqrt(1/2)

// (re,im) *= W16^K   (K in 0..7), with the trivial cases folded at compile time
template <int K>
ZD_HD void mul_w16(double &re, double &im) {
    if constexpr (K == 0) {
    } else if constexpr (K == 4) {
        double t = re;
        re       = -im;
        im       = t;
    } else if constexpr (K == 2) {
        double a = (re - im) * ZD_R2, b = (re + im) * ZD_R2;
        re = a;
        im = b;
    } else if constexpr (K == 6) {
        double a = (-re - im) * ZD_R2, b = (re - im) * ZD_R2;
        re = a;
        im = b;
    } else {
        constexpr double c = (K == 1) ? ZD_C1 : (K == 3) ? ZD_S1 : (K == 5) ? -ZD_S1 : -ZD_C1;
        constexpr double s = (K == 1) ? ZD_S1 : (K == 3) ? ZD_C1 : (K == 5) ? ZD_C1 : ZD_S1;
        double a = re * c - im * s, b = re * s + im * c;
        re = a;
        im = b;
    }
}

constexpr int bitrev(int i, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++)
        if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}

// One radix-2 DIF stage over the R-point butterfly living in slots BASE + STRIDE*n.
template <int R, int HALF, int BASE, int STRIDE, int E, int BLK = 0, int K = 0>
ZD_HD void dif_stage(double (&re)[E], double (&im)[E]) {
    if constexpr (BLK < R) {
        if constexpr (K < HALF) {
            constexpr int i0 = BASE + STRIDE * (BLK + K);
            constexpr int i1 = BASE + STRIDE * (BLK + K + HALF);
            double ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
            re[i0]    = ar + br;
            im[i0]    = ai + bi;
            double dr = ar - br, di = ai - bi;
            mul_w16<K * (8 / HALF)>(dr, di);  // W_{2 HALF}^K = W16^{K*16/(2 HALF)}
            re[i1] = dr;
            im[i1] = di;
            dif_stage<R, HALF, BASE, STRIDE, E, BLK, K + 1>(re, im);
        } else {
            dif_stage<R, HALF, BASE, STRIDE, E, BLK + 2 * HALF, 0>(re, im);
        }
    }
}

template <int R, int BASE, int STRIDE, int E, int I = 0>
ZD_HD void unrev(double (&re)[E], double (&im)[E], const double (&tr)[R], const double (&ti)[R]) {
    if constexpr (I < R) {
        re[BASE + STRIDE * I] = tr[bitrev(I, ilog2c(R))];
        im[BASE + STRIDE * I] = ti[bitrev(I, ilog2c(R))];
        unrev<R, BASE, STRIDE, E, I + 1>(re, im, tr, ti);
    }
}
template <int R, int BASE, int STRIDE, int E, int I = 0>
ZD_HD void gather(const double (&re)[E], const double (&im)[E], double (&tr)[R], double (&ti)[R]) {
    if constexpr (I < R) {
        tr[I] = re[BASE + STRIDE * I];
        ti[I] = im[BASE + STRIDE * I];
        gather<R, BASE, STRIDE, E, I + 1>(re, im, tr, ti);
    }
}

// R-point DFT (sign +1), natural order in and out, on slots BASE + STRIDE*n of the register arrays.
template <int R, int BASE, int STRIDE, int E>
ZD_HD void dft_inreg(double (&re)[E], double (&im)[E]) {
    static_assert(R == 1 || R == 2 || R == 4 || R == 8 || R == 16, "radix");
    if constexpr (R >= 16) dif_stage<R, 8, BASE, STRIDE, E>(re, im);
    if constexpr (R >= 8) dif_stage<R, 4, BASE, STRIDE, E>(re, im);
    if constexpr (R >= 4) dif_stage<R, 2, BASE, STRIDE, E>(re, im);
    if constexpr (R >= 2) dif_stage<R, 1, BASE, STRIDE, E>(re, im);
    if constexpr (R >= 4) {  // bit-reversed -> natural (pure register renaming once unrolled)
        double tr[R], ti[R];
        gather<R, BASE, STRIDE, E>(re, im, tr, ti);
        unrev<R, BASE, STRIDE, E>(re, im, tr, ti);
    }
}

template <int R, int B, int E, int Q = 0>
ZD_HD void dft_all(double (&re)[E], double (&im)[E]) {
    if constexpr (Q < B) {
        dft_inreg<R, Q, B, E>(re, im);
        dft_all<R, B, E, Q + 1>(re, im);
    }
}

// slot q + r*B  *=  w^r  for r = 1..R-1.  Powers are produced in r order and consumed at once so
// that only {w, w^2, latest even power, current} are live (registers are the scarce resource at
// E = 16): odd r: w^r = w^(r-1) * w, even r: w^r = w^(r-2) * w^2  (multiply depth <= R/2 + 1).
template <int R, int B, int E, int Rr>
ZD_HD void twiddle_chain(double (&re)[E], double (&im)[E], int q, double w1r, double w1i, double w2r, double w2i,
                         double evr, double evi) {
    if constexpr (Rr < R) {
        double cr, ci;
        if constexpr (Rr == 1) {
            cr = w1r;
            ci = w1i;
        } else if constexpr (Rr == 2) {
            cr = w2r;
            ci = w2i;
        } else if constexpr ((Rr & 1) == 0) {
            cr = evr * w2r - evi * w2i;
            ci = evr * w2i + evi * w2r;
        } else {
            cr = evr * w1r - evi * w1i;
            ci = evr * w1i + evi * w1r;
        }
        const int s    = q + Rr * B;
        const double a = re[s] * cr - im[s] * ci;
        const double b = re[s] * ci + im[s] * cr;
        re[s]          = a;
        im[s]          = b;
        if constexpr ((Rr & 1) == 0)
            twiddle_chain<R, B, E, Rr + 1>(re, im, q, w1r, w1i, w2r, w2i, cr, ci);
        else
            twiddle_chain<R, B, E, Rr + 1>(re, im, q, w1r, w1i, w2r, w2i, evr, evi);
    }
}

// Pass P of the Stockham schedule: twiddle (none in pass 0) + in-register butterflies.
//   tw: table exp(+2 pi i k / N), k < N  (16-byte entries, global memory / L1-resident)
template <class PL, int P>
ZD_HD void pass_compute(double (&re)[PL::E], double (&im)[PL::E], int t, const cplx *__restrict__ tw) {
    constexpr int E = PL::E, R = PL::radix(P), B = E / R, NS = PL::ns(P), T = PL::T, N = PL::N;
    if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int j   = t + q * T;
            const int m   = (j & (NS - 1)) * (N / (NS * R));
            const cplx w1 = tw[m];
            const double w2r = w1.x * w1.x - w1.y * w1.y, w2i = 2.0 * w1.x * w1.y;
            twiddle_chain<R, B, E, 1>(re, im, q, w1.x, w1.y, w2r, w2i, 1.0, 0.0);
        }
    }
    dft_all<R, B, E>(re, im);
}

// The inter-pass twiddles w1 of a thread depend on its index only: a caller that has to wait for its input data anyway
// can fetch them up front (TwSet / load_twiddles) and run fft_line with them, instead of two or three dependent loads
// from the table in the middle of the transform.
// (Only passes with at most two butterflies per thread are fetched ahead — the full-radix ones: a short last pass has E/2
// twiddles per thread, more registers than the wait is worth.)
template <class PL>
struct TwSet {
    static constexpr bool ahead(int p) { return PL::E / PL::radix(p) <= 2; }
    cplx v[PL::NPASS][2];  // [pass][q]; pass 0 has none
};
template <class PL, int P = 1>
ZD_HD void load_twiddles(TwSet<PL> &T, int t, const cplx *__restrict__ tw) {
    if constexpr (P < PL::NPASS) {
        constexpr int E = PL::E, R = PL::radix(P), B = E / R, NS = PL::ns(P), N = PL::N;
        if constexpr (TwSet<PL>::ahead(P)) {
#pragma unroll
            for (int q = 0; q < B; q++) T.v[P][q] = tw[((t + q * PL::T) & (NS - 1)) * (N / (NS * R))];
        }
        load_twiddles<PL, P + 1>(T, t, tw);
    }
}
template <class PL, int P>
ZD_HD void pass_compute_tw(double (&re)[PL::E], double (&im)[PL::E], int t, const cplx *__restrict__ tw, const TwSet<PL> &T) {
    constexpr int E = PL::E, R = PL::radix(P), B = E / R;
    if constexpr (P > 0 && !TwSet<PL>::ahead(P)) {
        pass_compute<PL, P>(re, im, t, tw);
    } else {
        if constexpr (P > 0) {
#pragma unroll
            for (int q = 0; q < B; q++) {
                const cplx w1 = T.v[P][q];
                const double w2r = w1.x * w1.x - w1.y * w1.y, w2i = 2.0 * w1.x * w1.y;
                twiddle_chain<R, B, E, 1>(re, im, q, w1.x, w1.y, w2r, w2i, 1.0, 0.0);
            }
        }
        dft_all<R, B, E>(re, im);
    }
}

// Where output r of butterfly j (pass P) lives in the natural-order line.
template <class PL, int P>
ZD_HD int out_index(int t, int q, int r) {
    constexpr int R = PL::radix(P), NS = PL::ns(P), T = PL::T;
    const int j = t + q * T;
    return (j / NS) * (NS * R) + (j & (NS - 1)) + r * NS;
}

// LDS addressing policies: idx(o, w) = position of element o of line w inside the tile.
//  ColsInner: lines interleaved ([o][w]) — lanes run along w (strided-axis passes: y and z)
//  LineInner: each line contiguous with one pad double per 16 ([w][o + o/16]) — lanes run along t
// Both are (piecewise) linear in o, which the exchange code exploits: ONE address per thread plus
// compile-time offsets (ds_read/ds_write immediates) instead of 16 separately computed addresses.
//   lin(delta): idx(o + delta, w) == idx(o, w) + step(delta) for every o
//   unit16:     idx(o + r, w) == idx(o, w) + ustep(r) when o % 16 == 0 and 0 <= r < 16
// ColsInner pads one group of W doubles per 16 elements (like LineInner's one double): in the exchange after pass 0 the
// lanes of one column write element 16 j + r for consecutive j, i.e. 16*W doubles apart — a multiple of the 256-byte bank
// row for every W >= 2, a 16-way bank conflict on every ds_write; with the pad the 16 lanes x W columns spread over all
// banks (measured: PPD=4096 y pass ... see profiles/r02_tuning_notes.md).
template <int N, int W>
struct ColsInner {
    static constexpr int SIZE = (N + N / 16) * W;
    static ZD_HD int idx(int o, int w) { return (o + (o >> 4)) * W + w; }
    static constexpr bool lin(int delta) { return delta % 16 == 0; }
    static constexpr int step(int delta) { return (delta + delta / 16) * W; }
    static constexpr int ustep(int r) { return r * W; }
};
template <int N, int W>
struct LineInner {
    static constexpr int PITCH = N + N / 16;
    static constexpr int SIZE  = PITCH * W;
    static ZD_HD int idx(int o, int w) { return w * PITCH + o + (o >> 4); }
    static constexpr bool lin(int delta) { return delta % 16 == 0; }
    static constexpr int step(int delta) { return delta + delta / 16; }
    static constexpr int ustep(int r) { return r; }
};

template <class PL, int P, class LDS>
ZD_HD void xchg_write(const double (&v)[PL::E], int t, int w, double *lds) {
    constexpr int E = PL::E, R = PL::radix(P), B = E / R, NS = PL::ns(P), T = PL::T;
#pragma unroll
    for (int q = 0; q < B; q++) {
        const int j    = t + q * T;
        const int o0   = (j / NS) * (NS * R) + (j & (NS - 1));  // output 0 of butterfly j
        const int base = LDS::idx(o0, w);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (LDS::lin(NS))
                lds[base + LDS::step(r * NS)] = v[q + r * B];
            else if constexpr (NS == 1 && R == 16)  // o0 = 16 j: a multiple of 16
                lds[base + LDS::ustep(r)] = v[q + r * B];
            else
                lds[LDS::idx(o0 + r * NS, w)] = v[q + r * B];
        }
    }
}
template <class PL, class LDS>
ZD_HD void xchg_read(double (&v)[PL::E], int t, int w, const double *lds) {
    constexpr int T = PL::T;
    if constexpr (LDS::lin(T)) {
        const int base = LDS::idx(t, w);
#pragma unroll
        for (int e = 0; e < PL::E; e++) v[e] = lds[base + LDS::step(T * e)];
    } else {
#pragma unroll
        for (int e = 0; e < PL::E; e++) v[e] = lds[LDS::idx(t + T * e, w)];
    }
}

#if defined(__HIPCC__)
// Full in-place transform of the line distributed over the workgroup; `lds` holds LDS::SIZE doubles.
// All threads of the workgroup must call it (it contains barriers).
template <class PL, class LDS, int P = 0>
__device__ __forceinline__ void fft_line(double (&re)[PL::E], double (&im)[PL::E], int t, int w,
                                         double *lds, const cplx *__restrict__ tw) {
    pass_compute<PL, P>(re, im, t, tw);
    if constexpr (P + 1 < PL::NPASS) {
        xchg_write<PL, P, LDS>(re, t, w, lds);
        __syncthreads();
        xchg_read<PL, LDS>(re, t, w, lds);
        __syncthreads();
        xchg_write<PL, P, LDS>(im, t, w, lds);
        __syncthreads();
        xchg_read<PL, LDS>(im, t, w, lds);
        __syncthreads();
        fft_line<PL, LDS, P + 1>(re, im, t, w, lds, tw);
    }
}
// the same with the twiddles already in registers (load_twiddles)
template <class PL, class LDS, int P = 0>
__device__ __forceinline__ void fft_line_tw(double (&re)[PL::E], double (&im)[PL::E], int t, int w, double *lds,
                                            const cplx *__restrict__ tw, const TwSet<PL> &T) {
    pass_compute_tw<PL, P>(re, im, t, tw, T);
    if constexpr (P + 1 < PL::NPASS) {
        xchg_write<PL, P, LDS>(re, t, w, lds);
        __syncthreads();
        xchg_read<PL, LDS>(re, t, w, lds);
        __syncthreads();
        xchg_write<PL, P, LDS>(im, t, w, lds);
        __syncthreads();
        xchg_read<PL, LDS>(im, t, w, lds);
        __syncthreads();
        fft_line_tw<PL, LDS, P + 1>(re, im, t, w, lds, tw, T);
    }
}
#endif

}  // namespace zdfft
