// zd_genmath.h — per-mode device arithmetic shared by the generator kernels (zd_kernels.hip: k_gen / k_genf; zd_kernels_fz.hip:
// the fused generator + z FFT of the packed PLT store): table forms of ln / exp / sincos / sqrt, PowerSpectrum::power from the
// LDS image, the per-axis part of the eigenmode lookup.  Moved out of zd_kernels.hip unchanged (round 5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zd_device.h"

namespace zdgen {
using zd::GenConst;

// i mod N for 0 <= i < 2^31 (N a power of two on the production path; PPD = 2^a 3^b takes the division)
__device__ __forceinline__ int modn(int N, int i) { return (N & (N - 1)) == 0 ? (i & (N - 1)) : (int) ((unsigned) i % (unsigned) N); }

// interp_eigmode + get_eigenmode (src/zeldovich.cpp:154-276); out = e_x,e_y,e_z (weighted), lambda.
// The per-axis part of the lookup (table index or lower/upper corner + fraction, incl. the "never
// interpolate across the +-Nyquist seam" rule :176-183 and the wrap :194-198) depends on one wavenumber
// only, so the generator hoists x (fixed per thread) and y (fixed per row) out of its mode loop.
struct EigAxis {
    int l, h;   // exact stride: l = table index, h unused; interpolation: lower / upper corner
    double f;   // fraction towards h
};
__device__ __forceinline__ EigAxis eig_axis(const GenConst &g, int ik) {  // ik = table-space index 0..N-1
    EigAxis a;
    const int N = g.N, ep = (int) g.eig_ppd;
    if (ep % N == 0) {
        a.l = ik * (ep / N);
        a.h = a.l;
        a.f = 0.0;
        return a;
    }
    const int halfppd = ep / 2 + 1, ppdhalf = ep / 2;
    double f = ((double) ep) / N * ik;
    if (f > ppdhalf && f < halfppd) f = floor(f + 1);
    a.l = (int) f;
    a.h = a.l + 1;
    if (a.h == ep) a.h = 0;
    a.f = f - a.l;
    return a;
}
__device__ __forceinline__ int eig_index_x(const GenConst &g, int kx) { return kx < 0 ? g.N + kx : kx; }
__device__ __forceinline__ int eig_index_z(const GenConst &g, int kz) {
    const int i = kz < 0 ? g.N + kz : kz;
    return i > g.N / 2 ? g.N - i : i;  // +k half-space of the rfft layout
}
// ------------------------------------------------------------------------------------------------
// k_genf arithmetic.  The generator is VALU-bound (2 x 128-bit LCG steps + Box-Muller + P(k) per mode,
// redone for every z-residue pass), and a wave's 64 modes have 64 unrelated |k|^2: gathering {P, 1/k^2}
// from the by-|k|^2 table costs 64 cache-line fills per wave-instruction and was measured to take as
// long as all the arithmetic together.  So k_genf evaluates everything from small LDS tables
// (GenfTab, ~22 KB, built on the host in long double):
//   * one_rand<2> (power_spectrum.cpp:284-308) is kept as the exact integer m = r + 1 (m = 0 <=> r = 2^64-1,
//     i.e. the value 1.0) and its correctly rounded double; the 2^-64 scale is folded into the callers;
//   * ln x: x = 2^e f, f in [sqrt(1/2), sqrt(2)), bin j of width 1/256 with c_j ~ 1/centre, ln f = -ln c_j +
//     log1p(f c_j - 1) (7 terms; the two bins around f = 1 have c = 1 so that ln stays relatively exact near 1);
//   * e^x = 2^k 2^(j/64) e^r, |r| <= ln2/128 (6 terms);
//   * cos/sin(2 pi theta): nearest of 512 tabulated directions + a 3-term rotation;
//   * P(k): SplineFunction::val with per-segment records {x_lo, 1/h, y_lo, y_hi, y2_lo h^2/6, y2_hi h^2/6}.
// Each is within 4e-16 (relative) of the correctly rounded result; cgauss<2> itself evaluates
// cos/sin(fl(2*M_PI*theta)), 4e-16 away from the exact angle.
struct GenfTab {  // offsets in doubles inside the LDS image (zd_capi.cpp: build_genf_table)
    static constexpr int SC = 0, LG = 1024, EX = 1392, SEG = 1456, GLUT = 1024;
    static constexpr int lut(int nseg) { return SEG + 6 * nseg; }
    static constexpr int size(int nseg) { return SEG + 6 * nseg + GLUT / 4; }
};

// a*b + c as the three-address VOP3 form.  Left to itself the compiler picks the two-address v_fmac_f64 for the
// Horner steps below and then copies every coefficient into the destination first (a quarter of the loop's vector
// instructions were such v_mov_b64; the asm form makes the generator ~7 % faster).
// HAZARD: the compiler's hazard recognizer does not look at the operands of asm statements.  gfx950 needs a wait
// state between a transcendental VALU op (v_rcp_f64, v_rsq_f64, ...) and the first read of its result; with an asm
// v_fma_f64 as that first reader the wait state was missing whenever the scheduler happened to put the two back to
// back (round 1: 1/k^2 wrong -> PLT displacements 15 % off, NaN in the packed ZA kernel; the parity tests caught both).
// Two guards, neither by convention:
//   * every transcendental builtin used next to asm FMAs goes through trans_rcp / trans_rsq below, which carry their
//     own wait state (an `s_nop 0` tied to the result register: one issue cycle per reciprocal, ~3 per mode);
//   * build() runs check_trans_hazard.py on the gfx950 ISA of this file: no instruction may read a TRANS result in
//     the next issue slot.
// -DZD_NO_FMA_ASM builds the plain-fma() version of everything (A/B parity runs).
__device__ __forceinline__ double trans_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    asm volatile("s_nop 0" : "+v"(r));
    return r;
}
__device__ __forceinline__ double trans_rsq(double d) {
    double r = __builtin_amdgcn_rsq(d);
    asm volatile("s_nop 0" : "+v"(r));
    return r;
}
__device__ __forceinline__ double fma3(double a, double b, double c) {
#ifdef ZD_NO_FMA_ASM
    return fma(a, b, c);
#endif
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ double fnma3(double a, double b, double c) {  // -a*b + c
#ifdef ZD_NO_FMA_ASM
    return fma(-a, b, c);
#endif
    double d;
    asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double fmas3(double a, double b, double c) {  // a*b - c
#ifdef ZD_NO_FMA_ASM
    return fma(a, b, -c);
#endif
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, -%3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ double u64_to_double(uint64_t m) {  // round-to-nearest: both halves are exact, one rounding in the fma
    return fma((double) (uint32_t) (m >> 32), 4294967296.0, (double) (uint32_t) m);
}

// ln(x * 2^-ebias) for a normal x > 0
__device__ __forceinline__ double flog(double x, int ebias, const double *T) {
    double f = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e    = __builtin_amdgcn_frexp_exp(x);
    const bool lo = f < 0.70710678118654752440;
    f = lo ? f + f : f;  // [sqrt(1/2), sqrt(2))
    e = lo ? e - 1 : e;
    const int j     = (int) (f * 256.0) - 181;
    const double2 t = reinterpret_cast<const double2 *>(T + GenfTab::LG)[j];  // {c_j, -ln c_j}
    const double r  = fma(f, t.x, -1.0);
    double p = fma3(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma3(r, p, 0.2);
    p = fma3(r, p, -0.25);
    p = fma3(r, p, 1.0 / 3.0);
    p = fma3(r, p, -0.5);
    p = fma3(r, p, 1.0);
    const double de = (double) (e - ebias);
    return fma3(de, 0.69314716756343842, fma3(de, 1.2996506893901347e-08, fma3(r, p, t.y)));
}

__device__ __forceinline__ double fexp(double x, const double *T) {
    x = fmin(fmax(x, -745.0), 709.0);
    const double n = __builtin_rint(x * 92.332482616893656877);  // 64 / ln 2
    double r = fma3(n, -0.010830424493178725, x);
    r = fma3(n, -2.0307042021720854e-10, r);
    const int ni = (int) n;
    double p = fma3(r, 1.0 / 720, 1.0 / 120);
    p = fma3(r, p, 1.0 / 24);
    p = fma3(r, p, 1.0 / 6);
    p = fma3(r, p, 0.5);
    p = fma3(r, p, 1.0);
    p = fma3(r, p, 1.0);
    return ldexp(T[GenfTab::EX + (ni & 63)] * p, ni >> 6);
}

__device__ __forceinline__ double frcp(double d) {
    double r = trans_rcp(d);
    const double e = fma(-d, r, 1.0);
    r = fma3(e, r, r);
    return fma3(fnma3(d, r, 1.0), r, r);
}

// sqrt(v) for v >= 0 (v = 0 -> 0); v is far from the subnormal range (P(k) |ln R| of a mode that carries power)
__device__ __forceinline__ double sqrt_pos(double v) {
    const double r = trans_rsq(v);
    double g = v * r, h = 0.5 * r;
    const double e = fnma3(h, g, 0.5);
    g = fma3(g, e, g);
    h = fma3(h, e, h);
    g = fma3(fnma3(g, g, v), h, g);
    g = fma3(fnma3(g, g, v), h, g);
    return v > 0.0 ? g : 0.0;
}

// cos/sin(2 pi m 2^-64) from sc[j] = {cos, sin}(2 pi j / 512)
__device__ __forceinline__ void sincos_u01(double md, const double *T, double &sn, double &cs) {
    const double t = md * 2.77555756156289135e-17;  // m * 2^-55 = theta * 512, exact
    const double j = __builtin_rint(t);
    const double b = (t - j) * 1.22718463030851298e-02;  // 2 pi / 512 * (theta*512 - j), |b| <= pi/512
    const double2 a = reinterpret_cast<const double2 *>(T + GenfTab::SC)[((int) j) & 511];
    const double b2 = b * b;
    const double sb = fma3(b * b2, fma3(b2, 8.33333333333333322e-03, -1.66666666666666657e-01), b);          // sin b
    const double cm = b2 * fma3(b2, fma3(b2, -1.38888888888888894e-03, 4.16666666666666644e-02), -0.5);     // cos b - 1
    cs = a.x + fmas3(a.x, cm, a.y * sb);
    sn = a.y + fma3(a.y, cm, a.x * sb);
}

// PowerSpectrum::power (src/power_spectrum.cpp:225-261) for k2 = |k|^2 > 0, from the LDS image
template <bool PLAW>
__device__ __forceinline__ double genf_power(const GenConst &g, const double *T, double k2) {
    const double v = 0.5 * flog(k2, 0, T);  // ln k
    double val;
    if constexpr (PLAW) {
        val = g.powerlaw_index * v;
    } else {
        const int nseg = g.genf_nseg;
        const unsigned short *lut = reinterpret_cast<const unsigned short *>(T + GenfTab::lut(nseg));
        int c = (int) ((v - g.glut_x0) * g.glut_inv_dx);
        c     = c < 0 ? 0 : (c >= GenfTab::GLUT ? GenfTab::GLUT - 1 : c);
        int klo = lut[c];
        const double *seg = T + GenfTab::SEG;
        while (klo < nseg - 1 && seg[6 * (klo + 1)] <= v) klo++;  // largest klo with x[klo] <= v (spline_function.h:146-152)
        const double2 *rec = reinterpret_cast<const double2 *>(seg + 6 * klo);
        const double2 q0 = rec[0], q1 = rec[1], q2 = rec[2];  // {x_lo, 1/h}, {y_lo, y_hi}, {c_lo, c_hi}
        const double b = (v - q0.x) * q0.y, a = 1.0 - b;
        val = fma3(a, q1.x, b * q1.y) + fma3(fmas3(a * a, a, a), q2.x, fmas3(b * b, b, b) * q2.y);
    }
    return fexp(fma(-k2, g.pk_smooth2, val), T) * g.pk_norm;
}

__device__ __forceinline__ void cmac(double &ar, double &ai, double c, double dr, double di) {
    ar = fma(c, dr, ar);
    ai = fma(c, di, ai);
}

}  // namespace zdgen
