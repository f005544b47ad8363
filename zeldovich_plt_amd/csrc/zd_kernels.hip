// zd_kernels.hip — the four gfx950 kernels of the grid->displacements path and their launchers.
//
//   k_gen    : Gaussian modes  D(k) = sqrt(-P(k) ln R) e^{2 pi i theta}  (+ PLT factors)
//              replaces the (z,x) loop of LoadPlane, src/zeldovich.cpp:333-438, cgauss<2>
//              (src/power_spectrum.cpp:338-359) and get_eigenmode (src/zeldovich.cpp:229-276)
//   k_zfft   : Hermitian packing + (folded) z FFT, replaces zeldovich.cpp:447-511 and StoreBlock
//   k_yfft   : y FFT in place on the block store, replaces LoadBlock + half of Inverse2dFFT
//   k_xfft   : x FFT + particle epilogue, replaces the rest of Inverse2dFFT + WriteParticlesSlab
//              (src/output.cpp:86-203)
// HBM-bound design notes are in DESIGN.md; LDS/register structure of the FFT in zd_fft.h.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "zd_device.h"
#include "zd_epi.h"
#include "zd_launch.h"
#include "zd_genmath.h"

using namespace zd;
using zdfft::cplx;
using zdpcg::u128;
using namespace zdgen;

__constant__ zdpcg::BitTable c_bits;

// streaming (non-temporal) 16-byte accesses for data that is touched once per pass: tuning knob ZD_NT
typedef double zd_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cplx ld_stream(const cplx *p, bool nt) {
    if (nt) {
        const zd_d2v q = __builtin_nontemporal_load(reinterpret_cast<const zd_d2v *>(p));
        return cplx{q.x, q.y};
    }
    return *p;
}
__device__ __forceinline__ void st_stream(cplx *p, cplx v, bool nt) {
    if (nt) {
        zd_d2v q;
        q.x = v.x;
        q.y = v.y;
        __builtin_nontemporal_store(q, reinterpret_cast<zd_d2v *>(p));
    } else
        *p = v;
}

// ZD_NTBIT(S, b): bit b of the ZD_NT knob — from the environment in the tuning library, or fixed at compile time
// (-DZD_NT_FORCE=bits on otherwise product flags: `make nt32`, ... for clean A/B timings; the tuning library's ablation branches
// cost the y kernel 65 spilled registers)
#ifdef ZD_NT_FORCE
#define ZD_NTBIT(S, b) (((ZD_NT_FORCE) & (b)) != 0)
#else
#define ZD_NTBIT(S, b) ZD_TUNE((S).nt & (b))
#endif

#ifdef ZD_STAMPS
// Diagnostic stamps (`make stamps`: the tuning library + -DZD_STAMPS — the stamps cost the y kernel 87 spilled registers, so the
// plain tuning library, whose timings are compared, leaves them out): wave 0 of every workgroup of the y stage records shader-clock times of its phases,
// the 100 MHz real-time clock at its start and the hardware id of its CU into zd_stamps[unit * 8 ...] (scripts/yf_stamps.py).
__device__ unsigned long long *zd_stamps = nullptr;
extern "C" int zdk_set_stamps(unsigned long long *buf) { return (int) hipMemcpyToSymbol(HIP_SYMBOL(zd_stamps), &buf, sizeof(buf)); }
#define ZD_STAMP(slot, unit, drain)                                                                        \
    do {                                                                                                   \
        if (zd_stamps) {                                                                                   \
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
            if (threadIdx.x == 0) zd_stamps[(size_t) (unit) * 8 + (slot)] = __builtin_amdgcn_s_memtime();   \
        }                                                                                                  \
    } while (0)
#else
#define ZD_STAMP(slot, unit, drain) \
    do {                            \
    } while (0)
#endif

extern "C" int zdk_upload_bit_table(const zdpcg::BitTable *host) {
    return (int) hipMemcpyToSymbol(HIP_SYMBOL(c_bits), host, sizeof(zdpcg::BitTable));
}

// ------------------------------------------------------------------------------------------------
// device math for one mode

__device__ __forceinline__ u128 advance_bits(u128 s, uint64_t delta) {
    for (int i = 0; i < zdpcg::NBITS; i++) {
        if ((delta >> i) == 0) break;
        if ((delta >> i) & 1ULL) s = zdpcg::apply(c_bits.m[i], s);
    }
    return s;
}

// folded inputs of the field store: Y[field j][row group][k2][x][8 rows] (k_zfft_fb reads 128-byte lines: one column, the
// 8 rows of a group)
__device__ __forceinline__ unsigned y_index_blocked(int j, int kyl, int nky, int L, int k2, int N, int x) {
    return (unsigned) (((((j * (nky / FIELD_RB) + kyl / FIELD_RB) * L + k2) * N + x) * FIELD_RB) + (kyl & (FIELD_RB - 1)));
}

// PowerSpectrum::power (src/power_spectrum.cpp:225-261) with SplineFunction::val
// (include/spline_function.h:141-163).  The reference bisects for the segment; here a uniform-cell
// table over ln k gives a start index and a short forward scan lands on exactly the segment the
// bisection would return (largest klo with x[klo] <= v, clamped to [0, n-2]).
template <bool PLAW>
__device__ __forceinline__ double pk_power(const GenConst &g, double k2) {  // k2 = |k|^2
    if (k2 <= 0.0) return 0.0;
    if constexpr (PLAW) {
        const double k = sqrt(k2);
        return pow(k, g.powerlaw_index) * exp(-k2 * g.pk_smooth2) * g.pk_norm;
    } else {
        const double v = 0.5 * log(k2);  // = log(sqrt(k2)) to 1 ulp
        int c   = (int) ((v - g.lut_x0) * g.lut_inv_dx);
        c       = c < 0 ? 0 : (c >= PK_LUT ? PK_LUT - 1 : c);
        int klo = g.pk_lut[c];
        const int last = g.pk_n - 2;
        while (klo < last && g.pk_x[klo + 1] <= v) klo++;
        const int khi = klo + 1;
        const double xl = g.pk_x[klo], xh = g.pk_x[khi];
        const double h = xh - xl;
        const double a = (xh - v) / h, b = (v - xl) / h;
        const double val = a * g.pk_y[klo] + b * g.pk_y[khi]
                           + ((a * a * a - a) * g.pk_y2[klo] + (b * b * b - b) * g.pk_y2[khi]) * (h * h) / 6.0;
        return exp(val - k2 * g.pk_smooth2) * g.pk_norm;
    }
}

// sin and cos of 2*pi*theta for theta in (0,1]: exact octant reduction (theta*8 is exact), then the
// fdlibm kernel polynomials on [0, pi/4].  cgauss<2> (power_spectrum.cpp:353-356) evaluates
// cos/sin(fl(2*M_PI*theta)); the two differ by the rounding of that product, ~4e-16 absolute.
__device__ __forceinline__ void sincos2pi(double theta, double &sn, double &cs) {
    const double t8 = theta * 8.0;
    const int q     = (int) t8;
    const double f  = t8 - (double) q;
    const int k     = q & 7;
    const double r  = (k & 1) ? 1.0 - f : f;
    const double a  = 0.78539816339744830962 * r;  // pi/4 * r
    const double z  = a * a;
    // __kernel_sin / __kernel_cos coefficients (fdlibm k_sin.c, k_cos.c)
    const double sp = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04
                      + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double s0 = a + a * z * sp;
    const double cp = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05
                      + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double c0 = 1.0 - (0.5 * z - z * z * cp);
    const bool swap = ((k + 1) & 2) != 0;  // k = 1,2,5,6
    const double sv = swap ? c0 : s0, cv = swap ? s0 : c0;
    sn = (k >= 4) ? -sv : sv;                      // k = 4..7: sin < 0
    cs = (k >= 2 && k <= 5) ? -cv : cv;            // k = 2..5: cos < 0
}

// cgauss<2> (power_spectrum.cpp:338-359) given P(k) and the two raw draws
__device__ __forceinline__ void gauss_from_pk(const GenConst &g, double Pk, uint64_t r1, uint64_t r2, double &dr,
                                              double &di) {
    double R           = zdpcg::u01(r1);
    const double theta = zdpcg::u01(r2);
    if ZD_TUNE(g.ablate & 2) {
        dr = R * Pk;
        di = theta * Pk;
        return;
    }
    if (!g.fixed_power)
        R = sqrt(-Pk * log(R));
    else
        R = sqrt(Pk);
    double sn, cs;
    sincos2pi(theta, sn, cs);
    dr = R * cs;
    di = R * sn;
}
template <bool PLAW>
__device__ __forceinline__ void gauss_mode(const GenConst &g, double k2, uint64_t r1, uint64_t r2, double &dr,
                                           double &di) {
    gauss_from_pk(g, ZD_TUNE(g.ablate & 1) ? 1e-9 * k2 : pk_power<PLAW>(g, k2), r1, r2, dr, di);
}

__device__ __forceinline__ void get_eigenmode_dev(const GenConst &g, int kx, int ky, int kz, const EigAxis &ax,
                                                  const EigAxis &ay, const EigAxis &az, double (&out)[4]) {
    const int ep = (int) g.eig_ppd, halfppd = ep / 2 + 1;
    const double k2 = (double) (kx * kx + ky * ky + kz * kz);
    double eh[4];
    const double2 *E = reinterpret_cast<const double2 *>(g.eig);  // [x][y][z][4] doubles = 2 double2 per entry
    if (ep % g.N == 0) {
        const int i = ((ax.l * ep + ay.l) * halfppd + az.l) * 2;
        const double2 q0 = E[i], q1 = E[i + 1];
        eh[0] = q0.x;
        eh[1] = q0.y;
        eh[2] = q1.x;
        eh[3] = q1.y;
    } else {
        // trilinear weights and corners in the reference's order f[0..7] = (x l/h, y l/h, z l/h) with z
        // fastest; accumulated left to right like the single expression of zeldovich.cpp:218-225.
        // Corners with zero weight are not read (the reference reads them: same value unless non-finite).
        eh[0] = eh[1] = eh[2] = eh[3] = 0.0;
#pragma unroll 2
        for (int c = 0; c < 8; c++) {
            const double wx = (c & 4) ? ax.f : 1 - ax.f, wy = (c & 2) ? ay.f : 1 - ay.f, wz = (c & 1) ? az.f : 1 - az.f;
            const double wgt = wx * wy * wz;
            if (wgt != 0) {
                const int cx = (c & 4) ? ax.h : ax.l, cy = (c & 2) ? ay.h : ay.l, cz = (c & 1) ? az.h : az.l;
                const int i  = ((cx * ep + cy) * halfppd + cz) * 2;
                const double2 q0 = E[i], q1 = E[i + 1];
                eh[0] += wgt * q0.x;
                eh[1] += wgt * q0.y;
                eh[2] += wgt * q1.x;
                eh[3] += wgt * q1.y;
            }
        }
    }
    eh[2] *= (kz < 0 ? -1.0 : 1.0);  // copysign(1, kz) for an int: kz = 0 -> +1
    const double imag = 1.0 / sqrt(eh[0] * eh[0] + eh[1] * eh[1] + eh[2] * eh[2]);
    eh[0] *= imag;  // the reference divides each component (zeldovich.cpp:257-259): <= 1 ulp apart
    eh[1] *= imag;
    eh[2] *= imag;
    double norm = k2 / (kx * eh[0] + ky * eh[1] + kz * eh[2]);
    if (k2 == 0.0 || !isfinite(norm)) norm = 0.0;
    out[0] = norm * eh[0];
    out[1] = norm * eh[1];
    out[2] = norm * eh[2];
    out[3] = eh[3];
}

// table of {P(k), 1/k^2} indexed by the integer kx^2+ky^2+kz^2 (built with the same device code that
// would evaluate it per mode, so table and direct evaluation agree bitwise)
template <bool PLAW>
__global__ void k_pk_table(GenConst g, int n, double2 *__restrict__ tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double k2 = (double) i * g.fundamental2;
    double2 v;
    v.x = pk_power<PLAW>(g, k2);
    if (k2 == 0.0) k2 = 1.0;
    v.y = 1.0 / k2;
    tab[i] = v;
}

// ------------------------------------------------------------------------------------------------
// k_gen: mode generation fused with the Hermitian job algebra and the z-residue fold.
//   One thread owns one x and ZR consecutive k2 (k2 < L = N/R).  For each k2 it visits the R modes
//   kz-index z = k2 + L*k1, draws D(k), forms the NJ job inputs c_j(k) D(k) and accumulates
//        Y_j[k2] = W_N^{k2 r} * sum_k1 W_R^{k1 r} c_j D            (r = residue)
//   i.e. the decimation-in-frequency fold, so that k_zfft only has to run length-L FFTs.
//   The RNG walk is z-major with two stride maps (forward L rows; back to the next k2), each with a
//   variant for crossing the z = N/2 wrap of the counter (zeldovich.cpp:335).
//   ky = 0: "loser" positions take the conjugate of the winner's mode (zeldovich.cpp:485-503); that
//   plane uses direct counter addressing per mode (1 of N/2 planes).
//   Y[((j*nky + kyl)*L + k2)*N + x]   complex
// grid: (ceil(N/GEN_BX), L/ZR, nky)  block: GEN_BX
template <int ZR, int NJ, bool PLT, bool PLAW>
__global__ __launch_bounds__(GEN_BX) void k_gen(GenConst g, GenJumps J, JobList jobs, StoreLayout S, int zW, int ky0,
                                                int nky, int L, int residue, int residue2,
                                                const cplx *__restrict__ twN, cplx *__restrict__ Y) {
    const int N = g.N, half = g.half, R = N / L;
    const int x   = blockIdx.x * GEN_BX + threadIdx.x;
    const int k20 = blockIdx.y * ZR;
    const int kyl = blockIdx.z;
    const int ky  = ky0 + kyl * S.ky_stride;
    if (x >= N) return;
    const int kx = x > half ? x - N : x;
    if (S.prune & 1) {  // the whole k_zfft tile this column belongs to is identically zero: nothing to produce
        // k_zfft reads this column through the tile [xt0, xt0+zW) ("self" jobs) and through the tile
        // shifted by one column ("twin" jobs): skip only if both are entirely zero
        bool all_zero = true;
        const int xt0 = x - x % zW;
        for (int i = -1; i <= zW; i++) {
            const int xi = modn(N, xt0 + i + N);
            all_zero = all_zero && column_is_zero(S, xi > half ? xi - N : xi, ky);
        }
        if (all_zero) return;
    }
    EigAxis eax = {0, 0, 0.0}, eay = {0, 0, 0.0};
    if constexpr (PLT) {
        eax = eig_axis(g, eig_index_x(g, kx));
        eay = eig_axis(g, ky);
    }
    double vsum = 0.0;  // sum |D|^2 over the positions this thread visits (packed stores only)
    u128 s = 0;
    if (ky != 0) {  // state one step ahead of the first mode's counter
        const int kz0 = k20 > half ? k20 - N : k20;  // k20 > N/2 only happens when R = 1
        const uint64_t off = 2ULL * ((uint64_t) (kz0 & 65535) * 65536ULL + (uint64_t) (kx & 65535)) + 1ULL;
        s = advance_bits(g.row_state[ky], off);
    }
#pragma unroll 1
    for (int zi = 0; zi < ZR; zi++) {
        const int k2 = k20 + zi;
        double accr[NJ], acci[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) accr[j] = acci[j] = 0.0;
#pragma unroll 1
        for (int k1 = 0; k1 < R; k1++) {
            const int z = k2 + L * k1;
            // ---- which mode feeds (ky, z, x), and its two raw draws ----
            int zs = z, xs = x;
            bool cj = false, zero = false;
            uint64_t r1, r2;
            if (ky != 0) {
                r1 = zdpcg::output(s);
                const u128 s2 = ZD_TUNE(g.ablate & 4) ? s + 12345 : zdpcg::step(s);
                r2 = zdpcg::output(s2);
                // next mode of the walk
                int zb;
                const zdpcg::Affine *m;
                if (k1 + 1 < R) {
                    zb = z + L;
                    m  = &J.fwd[(z > half) != (zb > half)];
                } else {
                    zb = k2 + 1;
                    m  = &J.back[(z > half) != (zb > half)];
                }
                s = ZD_TUNE(g.ablate & 4) ? s2 + m->C : zdpcg::apply(*m, s2);
            } else {
                if (z > half) {
                    zs = N - z;
                    xs = x ? N - x : 0;
                    cj = true;
                } else if (z == 0) {
                    if (x == 0)
                        zero = true;
                    else if (x > half) {
                        xs = N - x;
                        cj = true;
                    }
                }
                const int kxs = xs > half ? xs - N : xs, kzs = zs > half ? zs - N : zs;
                u128 t = advance_bits(g.row_state[0], 2ULL * ((uint64_t) (kzs & 65535) * 65536ULL + (uint64_t) (kxs & 65535)) + 1ULL);
                r1 = zdpcg::output(t);
                r2 = zdpcg::output(zdpcg::step(t));
            }
            const int kxm = xs > half ? xs - N : xs, kzm = zs > half ? zs - N : zs;  // generated mode
            const int k2i = kxm * kxm + ky * ky + kzm * kzm;
            double k2v    = (double) k2i * g.fundamental2;
            double dr = 0.0, di = 0.0, ik2 = 1.0;
            if (g.phik) {  // f_NL, second pass: D = phi_NG(k) * M(k) for every mode but k = 0 (zeldovich.cpp:393-400)
                if (k2i != 0) {
                    const cplx ph  = g.phik[((long long) (ky >> S.lG) * N + zs) * N + xs];  // PhiK rows are this rank's row slots
                    const double M = g.fnl_M[k2i];
                    dr  = ph.x * M;
                    di  = ph.y * M;
                    ik2 = 1.0 / k2v;
                }
            } else if (!zero && !mode_is_zero(g, kxm, ky, kzm, k2v) && g.v1dev) {
                // ZD_Version = 1, cgauss<1> (power_spectrum.cpp:310-332): the accepted pair of this mode, drawn by k_v1_draw
                const double2 ph = g.v1dev[((long long) kyl * N + zs) * N + xs];
                const double r2  = __dadd_rn(__dmul_rn(ph.x, ph.x), __dmul_rn(ph.y, ph.y));
                double Pk;
                if (g.pk_tab) {
                    const double2 pv = g.pk_tab[k2i];
                    Pk  = pv.x;
                    ik2 = pv.y;
                } else {
                    Pk  = pk_power<PLAW>(g, k2v);
                    ik2 = 1.0 / (k2v == 0.0 ? 1.0 : k2v);
                }
                const double q = g.fixed_power ? sqrt(Pk / r2) : sqrt(-Pk * log(r2) / r2);
                dr = ph.x * q;
                di = ph.y * q;
            } else if (!zero && !mode_is_zero(g, kxm, ky, kzm, k2v)) {
                if (g.pk_tab) {  // {P(k), 1/k^2} by integer k^2
                    const double2 pv = g.pk_tab[k2i];
                    ik2 = pv.y;
                    gauss_from_pk(g, ZD_TUNE(g.ablate & 1) ? 1e-9 * k2v : pv.x, r1, r2, dr, di);
                } else {
                    gauss_mode<PLAW>(g, k2v, r1, r2, dr, di);
                    ik2 = 1.0 / (k2v == 0.0 ? 1.0 : k2v);
                }
            }
            if (dr == 0.0 && di == 0.0) continue;  // zero modes add nothing (zeldovich.cpp:403,435-438)
            if (g.gen_phi) {  // f_NL, first pass: phi = D / M (zeldovich.cpp:385-391)
                const double M = g.fnl_M[k2i];
                dr /= M;
                di /= M;
            }
            double sx, sy, sz, f = 1.0;
            if constexpr (PLT) {
                double e[4];
                if ZD_TUNE(g.ablate & 8) {
                    e[0] = kxm; e[1] = ky; e[2] = kzm; e[3] = 1.0 - 1e-9 * k2i;
                } else {
                    const EigAxis eaz = eig_axis(g, eig_index_z(g, kzm));
                    if (cj) {  // mirrored source mode (ky = 0 plane only): its own x axis
                        const EigAxis eaxm = eig_axis(g, eig_index_x(g, kxm));
                        get_eigenmode_dev(g, kxm, ky, kzm, eaxm, eay, eaz, e);
                    } else
                        get_eigenmode_dev(g, kxm, ky, kzm, eax, eay, eaz, e);
                }
                f = (sqrt(1. + 24 * e[3] * g.f_cluster) - 1) * .25;
                double rescale = 1.0;
                if (g.qPLTrescale) rescale = exp(g.ln_growth_ratio * (g.target_f - f));
                sx = rescale * e[0] * g.fundamental * ik2;
                sy = rescale * e[1] * g.fundamental * ik2;
                sz = rescale * e[2] * g.fundamental * ik2;
            } else {
                sx = (double) kxm * g.fundamental * ik2;
                sy = (double) ky * g.fundamental * ik2;
                sz = (double) kzm * g.fundamental * ik2;
            }
            if (cj) {  // conjugated copy of the mode at -k: D -> conj D, s -> -s(-k)
                di = -di;
                sx = -sx;
                sy = -sy;
                sz = -sz;
            }
            vsum += dr * dr + di * di;
            const double se = g.fundamental * ik2;  // even in k: the conjugated copy keeps its sign (JOB_E)
            double d2r = dr, d2i = di;  // PACK_ZAPAIR / PACK_ZAFIELD: the same mode folded for the second residue of the pass
            if (R > 1) {  // W_R^{k1 r}
                const cplx w = twN[modn(N, k1 * residue * L)];
                const double a = dr * w.x - di * w.y, b = dr * w.y + di * w.x;
                if (jobs.pack == PACK_ZAPAIR || jobs.pack == PACK_ZAFIELD) {
                    const cplx w2 = twN[modn(N, k1 * residue2 * L)];
                    const double a2 = dr * w2.x - di * w2.y, b2 = dr * w2.y + di * w2.x;
                    d2r = a2;
                    d2i = b2;
                }
                dr = a;
                di = b;
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                double cr, ci;
                switch (jobs.kind[j]) {
                    case JOB_A_SELF: cr = 1.0 - sx; ci = 0.0; break;
                    case JOB_A_TWIN: cr = 1.0 + sx; ci = 0.0; break;
                    case JOB_B_SELF: cr = -sz; ci = sy; break;
                    case JOB_B_TWIN: cr = sz; ci = sy; break;
                    case JOB_C_BOTH: cr = -f * sx; ci = 0.0; break;
                    case JOB_D_SELF: cr = -f * sz; ci = f * sy; break;
                    case JOB_D_TWIN: cr = f * sz; ci = f * sy; break;
                    case JOB_XV_SELF: cr = -f * sx; ci = sx; break;
                    case JOB_XV_TWIN: cr = f * sx; ci = sx; break;
                    case JOB_FX: cr = 0.0; ci = sx; break;
                    case JOB_E: cr = se; ci = 0.0; break;
                    case JOB_Z: cr = sz; ci = 0.0; break;
                    case JOB_PX: cr = sx; ci = 0.0; break;
                    case JOB_PY: cr = sy; ci = 0.0; break;
                    case JOB_PZ: cr = sz; ci = 0.0; break;
                    case JOB_PFX: cr = f * sx; ci = 0.0; break;
                    case JOB_PFY: cr = f * sy; ci = 0.0; break;
                    case JOB_PFZ: cr = f * sz; ci = 0.0; break;
                    default: cr = 1.0; ci = 0.0; break;
                }
                const double ur = jobs.res[j] ? d2r : dr, ui = jobs.res[j] ? d2i : di;
                accr[j] += cr * ur - ci * ui;
                acci[j] += cr * ui + ci * ur;
            }
        }
        double pr = 1.0, pi = 0.0, qr = 1.0, qi = 0.0;  // W_N^{k2 r}, and the same for the second residue
        if (R > 1) {
            const cplx w = twN[modn(N, k2 * residue)];
            pr = w.x;
            pi = w.y;
            const cplx w2 = twN[modn(N, k2 * residue2)];
            qr = w2.x;
            qi = w2.y;
        }
        double outr[NJ], outi[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const double tr = jobs.res[j] ? qr : pr, ti = jobs.res[j] ? qi : pi;
            outr[j] = accr[j] * tr - acci[j] * ti;
            outi[j] = accr[j] * ti + acci[j] * tr;
        }
        if constexpr (NJ == 6) {
            if (jobs.pack == PACK_ZAPAIR) {  // jobs 4, 5 hold F_x of residue 0 / 1: X_self = F0 + i F1, X_twin input = F0 - i F1
                const double f0r = outr[4], f0i = outi[4], f1r = outr[5], f1i = outi[5];
                outr[4] = f0r - f1i;
                outi[4] = f0i + f1r;
                outr[5] = f0r + f1i;
                outi[5] = f0i - f1r;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            // a slab holds < 2^31 elements (1.5 GB / 16 B): 32-bit index arithmetic
            const unsigned idx = pack_is_fields(jobs.pack) ? y_index_blocked(j, kyl, nky, L, k2, N, x) : (unsigned) (((j * nky + kyl) * L + k2) * N + x);
            Y[idx] = cplx{outr[j], outi[j]};
        }
    }
    // full-space sum: a half-space row ky >= 1 stands for itself and its Hermitian twin.  (Per-thread atomics: with
    // k_genf active this kernel only sees the ky = 0 plane.)
    if (g.accum_var && vsum != 0.0) atomicAdd(&g.var_slots[(threadIdx.x + 5 * blockIdx.y) % NSLOT], vsum * (ky != 0 ? 2.0 : 1.0));
}

// ------------------------------------------------------------------------------------------------
// k_genf: the production generator for the half-space rows ky >= 1 (k_gen above stays the general
// kernel: the ky = 0 plane with its conjugate "loser" modes, the f_NL passes, the one-mode filter, the
// direct P(k) evaluation and the tuning ablations).  Same mapping, same RNG walk and same output as
// k_gen, restructured for the hardware:
//   * z (hence kz, the walk map, the fold twiddle and the |kz| = kmax rule) is wave-uniform: all of
//     that lives in SGPRs; the zero rule of zeldovich.cpp:350-353 is one integer compare per lane
//     (k2i_cut = smallest integer |k|^2 whose double product with fundamental^2 reaches k2_cutoff);
//   * no global-memory gathers: P(k), ln, exp and the Box-Muller direction come from LDS tables (above);
//   * a wave whose 64 modes are all zeroed only moves the RNG (ONE multiply-add by the full-stride map);
//   * the displacement algebra is accumulated per field, not per job: with s_j = k_j fund / k^2 and
//     kx, ky fixed per thread,  sum (1 -+ s_x) D w = S0 -+ kx SE,  sum (-+s_z + i s_y) D w = -+SZ + i ky SE
//     where S0 = sum D w, SE = sum (fund/k^2) D w, SZ = sum kz (fund/k^2) D w   (3 accumulators for the
//     4 ZA jobs; PLT keeps 7 because its eigenvectors differ mode by mode).
// grid: (ceil(N/GEN_BX), L/ZR, nrows)  block: GEN_BX          row kyl = kyl0 + blockIdx.z of the slab
enum { GENF_DENS = 0, GENF_ZA = 1, GENF_PLT = 2, GENF_ZAP = 3 /* PACK_ZAPAIR */, GENF_PLTN = 4 /* PACK_PLT3 */,
       GENF_ZAF = 5 /* PACK_ZAFIELD: the sums of GENF_ZAP written out as they are (E, Z of both residues) */,
       GENF_PLTF = 6 /* PACK_PLTFIELD: the sums of GENF_PLTN written out as they are */,
       GENF_ZAFD = 7 /* PACK_ZAFIELD + ZD_qdensity = 1: the four potentials and the density sum D of both residues (six fields) */ };

// Tail of the k_genf eigenmode lookups: from the blended (un-normalised) table entry eh = (e_x, e_y, e_z, lambda) of |kz| to the
// coefficient of the displacement, out[j] = s_j = e_j k^2 / (k.e) x fundamental / k^2 = e_j / ((k.e) fundamental) (round 5: neither
// |e| nor k^2 enters — the reference normalises e, src/zeldovich.cpp:255-262, forms e k^2 / (k.e) and LoadPlane multiplies by
// fundamental / k^2, :428-434; one reciprocal instead of an inverse square root with two Newton steps and two reciprocals: 25 of a
// PLT mode's ~350 instructions); k.e = 0 or not finite -> 0 as there (:262-263; k = 0 does not occur: ky >= 1); out[3] = lambda
__device__ __forceinline__ void eig_coeff_tail(double fundamental, int kx, int ky, int kz, double (&eh)[4], double (&out)[4]) {
    eh[2] *= (kz < 0 ? -1.0 : 1.0);
    const double dot = kx * eh[0] + ky * eh[1] + kz * eh[2];
    double inv = frcp(dot * fundamental);
    if (!isfinite(inv)) inv = 0.0;
    out[0] = inv * eh[0];
    out[1] = inv * eh[1];
    out[2] = inv * eh[2];
    out[3] = eh[3];
}

// get_eigenmode for k_genf: the (x, y) part of the lookup — table offsets of the 4 corner columns and the products
// w_x w_y — depends on the thread's kx and the row's ky only and is prepared once per tile; per mode remain the two
// z corners, 1/|e| by rsq + Newton and k^2/(k.e) by reciprocal (get_eigenmode_dev: sqrt + two divisions).
// Accumulation order and weight association are those of get_eigenmode_dev / zeldovich.cpp:218-225.
struct EigXY {
    int base[4];   // ((cx*ep + cy)*halfppd)*2 in double2 units, corners (l,l), (l,h), (h,l), (h,h)
    double w[4];   // w_x * w_y
};
__device__ __forceinline__ EigXY eig_xy(const GenConst &g, const EigAxis &ax, const EigAxis &ay) {
    const int ep = (int) g.eig_ppd, halfppd = ep / 2 + 1;
    EigXY q;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int cx = (c & 2) ? ax.h : ax.l, cy = (c & 1) ? ay.h : ay.l;
        q.base[c] = ((cx * ep + cy) * halfppd) * 2;
        q.w[c]    = ((c & 2) ? ax.f : 1 - ax.f) * ((c & 1) ? ay.f : 1 - ay.f);
    }
    return q;
}
__device__ __forceinline__ void eigenmode_fast(const GenConst &g, int kx, int ky, int kz, const EigXY &q, const EigAxis &az,
                                               double (&out)[4]) {
    const double fundamental = g.fundamental;
    const double2 *E = reinterpret_cast<const double2 *>(g.eig);
    double eh[4];
    if ((int) g.eig_ppd % g.N == 0) {
        const int i = q.base[0] + az.l * 2;
        const double2 q0 = E[i], q1 = E[i + 1];
        eh[0] = q0.x;
        eh[1] = q0.y;
        eh[2] = q1.x;
        eh[3] = q1.y;
    } else {
        eh[0] = eh[1] = eh[2] = eh[3] = 0.0;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const double wgt = q.w[c >> 1] * ((c & 1) ? az.f : 1 - az.f);
            if (wgt != 0) {
                const int i = q.base[c >> 1] + ((c & 1) ? az.h : az.l) * 2;
                const double2 q0 = E[i], q1 = E[i + 1];
                eh[0] += wgt * q0.x;
                eh[1] += wgt * q0.y;
                eh[2] += wgt * q1.x;
                eh[3] += wgt * q1.y;
            }
        }
    }
    eig_coeff_tail(fundamental, kx, ky, kz, eh, out);
}

// The (x, y) part of the trilinear lookup does not depend on kz: k_eig_lines does it once per column of a slab and pass — for
// every table cell cz along z the four (x, y) corners blended with their weights (zero-weight corners not read, like
// get_eigenmode_dev) — into lines[(cz * nrows + row) * N + x].  kz is wave-uniform in k_genf and consecutive lanes are
// consecutive x, so a mode reads two fully coalesced 32-byte entries per lane instead of 8 corners x 2 loads, and blends 4
// values with 2 weights instead of 32 FMAs with 8.  (Association differs from the single expression of zeldovich.cpp:218-225 —
// (w_x w_y) sums first, then w_z — by a few 1e-16 of the components.)  One slab is 65 x rows x N x 32 B (128 MB at PPD = 4096).
//   grid: (ceil(N / 256), nrows, eig_ppd / 2 + 1)   block: 256
// (one table cell cz per workgroup: with the cells in a loop the launch was 128 workgroups of 65 dependent table reads, 165 us in
// front of every generator launch — 42 ms per step at PPD=2048, 0.7 s at PPD=4096)
__global__ __launch_bounds__(256) void k_eig_lines(GenConst g, int ky0, int ky_stride, int nrows, double *__restrict__ lines) {
    const int N = g.N, x = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, cz = blockIdx.z;
    if (x >= N) return;
    const int kx = x > g.half ? x - N : x, ky = ky0 + row * ky_stride;
    const EigXY q = eig_xy(g, eig_axis(g, eig_index_x(g, kx)), eig_axis(g, ky));
    const double2 *E = reinterpret_cast<const double2 *>(g.eig);
    double2 *out = reinterpret_cast<double2 *>(lines);
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (q.w[c] != 0) {
            const double2 q0 = E[q.base[c] + cz * 2], q1 = E[q.base[c] + cz * 2 + 1];
            e0 += q.w[c] * q0.x;
            e1 += q.w[c] * q0.y;
            e2 += q.w[c] * q1.x;
            e3 += q.w[c] * q1.y;
        }
    const size_t i = (((size_t) cz * nrows + row) * N + x) * 2;
    out[i]     = double2{e0, e1};
    out[i + 1] = double2{e2, e3};
}

// the two (x, y)-interpolated entries a mode blends along z, as loaded (genf_tile_kz requests the NEXT fold term's entries before it
// works on the current one: at two waves per SIMD a PLT generator otherwise waits for these loads in every iteration)
struct EigRaw {
    double2 a0, a1, b0, b1;
    double wl, wh;
};
__device__ __forceinline__ EigRaw eig_lines_load(const GenConst &g, int kz, int row, int x) {
    const EigAxis az = eig_axis(g, eig_index_z(g, kz));
    const double2 *V = reinterpret_cast<const double2 *>(g.eig_lines);
    EigRaw r;
    r.wl = 1 - az.f;
    r.wh = az.f;
    r.a0 = r.a1 = r.b0 = r.b1 = double2{0.0, 0.0};
    if (r.wl != 0) {
        const size_t il = (((size_t) az.l * g.eig_rows + row) * g.N + x) * 2;
        r.a0 = V[il];
        r.a1 = V[il + 1];
    }
    if (r.wh != 0) {  // wave-uniform: kz is
        const size_t ih = (((size_t) az.h * g.eig_rows + row) * g.N + x) * 2;
        r.b0 = V[ih];
        r.b1 = V[ih + 1];
    }
    return r;
}
// eigenmode_lines from loaded entries (same arithmetic, same order)
__device__ __forceinline__ void eig_lines_finish(double fundamental, int kx, int ky, int kz, const EigRaw &r, double (&out)[4]) {
    double eh[4] = {0.0, 0.0, 0.0, 0.0};
    if (r.wl != 0) {
        eh[0] = r.wl * r.a0.x;
        eh[1] = r.wl * r.a0.y;
        eh[2] = r.wl * r.a1.x;
        eh[3] = r.wl * r.a1.y;
    }
    if (r.wh != 0) {
        eh[0] += r.wh * r.b0.x;
        eh[1] += r.wh * r.b0.y;
        eh[2] += r.wh * r.b1.x;
        eh[3] += r.wh * r.b1.y;
    }
    eig_coeff_tail(fundamental, kx, ky, kz, eh, out);
}

// eigenmode_fast from the slab's (x, y)-interpolated lines (k_eig_lines); `row` = row of the slab
__device__ __forceinline__ void eigenmode_lines(const GenConst &g, int kx, int ky, int kz, int row, int x, const EigAxis &az, double (&out)[4]) {
    const double fundamental = g.fundamental;
    const double2 *V = reinterpret_cast<const double2 *>(g.eig_lines);
    const size_t il = (((size_t) az.l * g.eig_rows + row) * g.N + x) * 2;
    const double wl = 1 - az.f, wh = az.f;
    double eh[4] = {0.0, 0.0, 0.0, 0.0};
    if (wl != 0) {
        const double2 a0 = V[il], a1 = V[il + 1];
        eh[0] = wl * a0.x;
        eh[1] = wl * a0.y;
        eh[2] = wl * a1.x;
        eh[3] = wl * a1.y;
    }
    if (wh != 0) {  // wave-uniform: kz is
        const size_t ih = (((size_t) az.h * g.eig_rows + row) * g.N + x) * 2;
        const double2 b0 = V[ih], b1 = V[ih + 1];
        eh[0] += wh * b0.x;
        eh[1] += wh * b0.y;
        eh[2] += wh * b1.x;
        eh[3] += wh * b1.y;
    }
    eig_coeff_tail(fundamental, kx, ky, kz, eh, out);
}

template <int ZR, int KIND, bool PLAW>
__device__ __forceinline__ double genf_tile(const GenConst &g, const GenJumps &J, const StoreLayout &S, const double *T, int zW,
                                            int ky0, int kyl_arg, int kyl_first, int nky, int L, int residue, int residue2, int bx,
                                            int by, const cplx *__restrict__ twN, cplx *__restrict__ Y) {
    // field sums per k2:  DENS {S0}  ZA {S0, SE, SZ}  PLT {S0, X, Y, Z, fX, fY, fZ}  ZAP {SE, SZ}(r0), {SE, SZ}(r1)
    //                     PLTN {X, Y, Z, fX, fY, fZ}
    constexpr bool ZA2 = KIND == GENF_ZAP || KIND == GENF_ZAF;  // two residues per pass
    constexpr int NACC = KIND == GENF_DENS ? 1 : KIND == GENF_ZA ? 3 : KIND == GENF_PLT ? 7 : ZA2 ? 4 : 6;
    constexpr bool IS_PLT = KIND == GENF_PLT || KIND == GENF_PLTN || KIND == GENF_PLTF;
    // field stores: a workgroup takes the 8 rows of a row GROUP (kyl_arg) x 32 columns, see genf_tile_mirror
    constexpr bool BLK = KIND == GENF_PLTF || KIND == GENF_ZAF;
    double vsum = 0.0;  // sum |D|^2 of this thread's modes (packed stores: density_variance by Parseval)
    const int N = g.N, half = g.half, R = N / L;
    const int rsub = BLK ? (int) (threadIdx.x & (FIELD_RB - 1)) : 0;
    const int x    = BLK ? bx * (GEN_BX / FIELD_RB) + (int) (threadIdx.x / FIELD_RB) : bx * GEN_BX + (int) threadIdx.x;
    const int kyl  = BLK ? kyl_arg * FIELD_RB + rsub : kyl_arg;
    const int k20 = by * ZR;
    const int ky  = ky0 + kyl * S.ky_stride;  // >= 1
    if (x >= N || kyl < kyl_first) return 0.0;
    const int kx = x > half ? x - N : x;
    if (S.prune & 1) {  // see k_gen: skip columns whose z-FFT tiles are all zero
        bool all_zero = true;
        const int xt0 = x - x % zW;
        if constexpr (BLK) {  // k_zfft_f tile: zW columns x the 8 rows of the group
            for (int i = 0; i < zW; i++)
                for (int r = 0; r < FIELD_RB; r++)
                    all_zero = all_zero && column_is_zero(S, (xt0 + i) > half ? xt0 + i - N : xt0 + i,
                                                          ky0 + (kyl_arg * FIELD_RB + r) * S.ky_stride);
        } else {
            for (int i = -1; i <= zW; i++) {
                const int xi = modn(N, xt0 + i + N);
                all_zero = all_zero && column_is_zero(S, xi > half ? xi - N : xi, ky);
            }
        }
        if (all_zero) return 0.0;
    }
    const int kxy2  = kx * kx + ky * ky;
    const bool dead = (kx < 0 ? -kx : kx) == g.kmax || ky == g.kmax;  // zeldovich.cpp:350
    EigXY exy = {};
    if constexpr (IS_PLT) {
        if (!g.eig_lines) exy = eig_xy(g, eig_axis(g, eig_index_x(g, kx)), eig_axis(g, ky));
    }
    u128 s;
    {  // state one step ahead of the first mode's counter
        const int kz0 = k20 > half ? k20 - N : k20;  // k20 > N/2 only happens when R = 1
        const uint64_t off = 2ULL * ((uint64_t) (kz0 & 65535) * 65536ULL + (uint64_t) (kx & 65535)) + 1ULL;
        s = advance_bits(g.row_state[ky], off);
    }
    const zdpcg::Affine fwd0 = J.fwd[0], fwdf0 = J.fwd_full[0];
#pragma unroll 1
    for (int zi = 0; zi < ZR; zi++) {
        const int k2 = k20 + zi;
        double accr[NACC], acci[NACC];
#pragma unroll
        for (int j = 0; j < NACC; j++) accr[j] = acci[j] = 0.0;
#pragma unroll 1
        for (int k1 = 0; k1 < R; k1++) {
            const int z  = k2 + L * k1;
            const int kz = z > half ? z - N : z;
            const int k2i = kxy2 + kz * kz;
            const bool live = !dead && (kz < 0 ? -kz : kz) != g.kmax && (g.corner_modes || k2i < g.k2i_cut);
            // the walk's next move: L rows on inside the fold (the common case keeps its map in SGPRs), else back to
            // the first term of the next k2; index 1 = the move crosses the z = N/2 wrap of the counter
            const bool last = k1 + 1 >= R;
            const int sel   = (z > half) != ((last ? k2 + 1 : z + L) > half);
            const bool any  = __any(live);
            zdpcg::Affine m;
            if (!last && !sel)
                m = any ? fwd0 : fwdf0;
            else if (!last)
                m = any ? J.fwd[1] : J.fwd_full[1];
            else
                m = any ? J.back[sel] : J.back_full[sel];
            if (!any) {  // all 64 modes zeroed: their draws are not needed, only the walk moves on
                s = zdpcg::apply(m, s);
                continue;
            }
            const uint64_t r1 = zdpcg::output(s);
            const u128 s2     = zdpcg::step(s);
            const uint64_t r2 = zdpcg::output(s2);
            s = zdpcg::apply(m, s2);
            // ---- cgauss<2> (power_spectrum.cpp:338-359); zeroed lanes of a live wave ride along with amplitude 0 ----
            const double k2v = (double) k2i * g.fundamental2;
            const double P   = ZD_TUNE(g.ablate & 16) ? 1e-9 * k2v : genf_power<PLAW>(g, T, k2v);  // bit 4: tuning ablation
            const double ik2 = frcp(k2v);
            const uint64_t m1 = r1 + 1ULL;  // one_rand<2>: u = (r + 1) 2^-64, and 1.0 for r = 2^64 - 1 (m = 0)
            double v = P;
            if (!g.fixed_power) v = -P * flog(u64_to_double(m1), 64, T);
            v = (m1 == 0 && !g.fixed_power) || !live ? 0.0 : v;
            const double amp = sqrt_pos(v);
            double sn, cs;
            sincos_u01(u64_to_double(r2 + 1ULL), T, sn, cs);  // m = 0 (theta = 1) lands on direction 0 like theta = 0
            double dr = amp * cs, di = amp * sn;
            vsum = fma(dr, dr, fma(di, di, vsum));
            double d2r = dr, d2i = di;  // ZAP: the mode folded for the second residue of the pass
            if (R > 1) {  // W_R^{k1 r}
                const cplx w = twN[modn(N, k1 * residue * L)];
                const double a = dr * w.x - di * w.y, b = dr * w.y + di * w.x;
                if constexpr (ZA2) {
                    const cplx w2 = twN[modn(N, k1 * residue2 * L)];
                    d2r = dr * w2.x - di * w2.y;
                    d2i = dr * w2.y + di * w2.x;
                }
                dr = a;
                di = b;
            }
            if constexpr (KIND == GENF_DENS || KIND == GENF_ZA || KIND == GENF_PLT) {
                accr[0] += dr;
                acci[0] += di;
            }
            if constexpr (KIND == GENF_ZA) {
                const double q  = g.fundamental * ik2;
                const double er = q * dr, ei = q * di;
                accr[1] += er;
                acci[1] += ei;
                cmac(accr[2], acci[2], (double) kz, er, ei);
            } else if constexpr (ZA2) {
                const double q  = g.fundamental * ik2, dkz = (double) kz;
                const double er = q * dr, ei = q * di, e2r = q * d2r, e2i = q * d2i;
                accr[0] += er;
                acci[0] += ei;
                cmac(accr[1], acci[1], dkz, er, ei);
                accr[2] += e2r;
                acci[2] += e2i;
                cmac(accr[3], acci[3], dkz, e2r, e2i);
            } else if constexpr (IS_PLT) {
                constexpr int B = KIND == GENF_PLT ? 1 : 0;  // index of the X sum (PLTN, PLTF: the six sums only)
                double e[4];
                if (g.eig_lines)  // (uniform) the slab's (x, y)-interpolated lines
                    eigenmode_lines(g, kx, ky, kz, kyl, x, eig_axis(g, eig_index_z(g, kz)), e);
                else
                    eigenmode_fast(g, kx, ky, kz, exy, eig_axis(g, eig_index_z(g, kz)), e);
                const double f = (sqrt_pos(1. + 24 * e[3] * g.f_cluster) - 1) * .25;
                double rescale = 1.0;
                if (g.qPLTrescale) rescale = fexp(g.ln_growth_ratio * (g.target_f - f), T);
                const double sx = rescale * e[0], sy = rescale * e[1], sz = rescale * e[2];  // e = the coefficients s_j (eig_coeff_tail)
                cmac(accr[B + 0], acci[B + 0], sx, dr, di);
                cmac(accr[B + 1], acci[B + 1], sy, dr, di);
                cmac(accr[B + 2], acci[B + 2], sz, dr, di);
                cmac(accr[B + 3], acci[B + 3], f * sx, dr, di);
                cmac(accr[B + 4], acci[B + 4], f * sy, dr, di);
                cmac(accr[B + 5], acci[B + 5], f * sz, dr, di);
            }
        }
        // ---- job inputs from the field sums, times W_N^{k2 r}; Y[((j*nky + kyl)*L + k2)*N + x] ----
        double pr = 1.0, pi = 0.0, qr = 1.0, qi = 0.0;
        if (R > 1) {
            const cplx w = twN[modn(N, k2 * residue)];
            pr = w.x;
            pi = w.y;
            if constexpr (ZA2) {
                const cplx w2 = twN[modn(N, k2 * residue2)];
                qr = w2.x;
                qi = w2.y;
            }
        }
        auto put = [&](int j, double vr, double vi) {
            const unsigned idx = BLK ? y_index_blocked(j, kyl, nky, L, k2, N, x) : (unsigned) (((j * nky + kyl) * L + k2) * N + x);
            Y[idx] = cplx{vr, vi};
        };
        auto putp = [&](int j, double vr, double vi) { put(j, vr * pr - vi * pi, vr * pi + vi * pr); };
        if constexpr (KIND == GENF_DENS) {
            putp(0, accr[0], acci[0]);
        } else if constexpr (KIND == GENF_ZAF) {
            auto putq = [&](int j, double vr, double vi) { put(j, vr * qr - vi * qi, vr * qi + vi * qr); };
            putp(0, accr[0], acci[0]);  // E(r0)
            putp(1, accr[1], acci[1]);  // Z(r0)
            putq(2, accr[2], acci[2]);  // E(r1)
            putq(3, accr[3], acci[3]);  // Z(r1)
        } else if constexpr (KIND == GENF_ZAP) {
            const double dkx = (double) kx, dky = (double) ky;
            // residue r0: B jobs and F_x = i kx SE, all times W_N^{k2 r0}
            const double y0r = dky * accr[0], y0i = dky * acci[0], y1r = dky * accr[2], y1i = dky * acci[2];
            putp(0, -accr[1] - y0i, -acci[1] + y0r);  // JOB_B_SELF (r0)
            putp(1, accr[1] - y0i, acci[1] + y0r);    // JOB_B_TWIN (r0)
            auto putq = [&](int j, double vr, double vi) { put(j, vr * qr - vi * qi, vr * qi + vi * qr); };
            putq(2, -accr[3] - y1i, -acci[3] + y1r);  // JOB_B_SELF (r1)
            putq(3, accr[3] - y1i, acci[3] + y1r);    // JOB_B_TWIN (r1)
            const double g0r = -dkx * acci[0], g0i = dkx * accr[0], g1r = -dkx * acci[2], g1i = dkx * accr[2];
            const double f0r = g0r * pr - g0i * pi, f0i = g0r * pi + g0i * pr;  // F_x(r0) W_N^{k2 r0}
            const double f1r = g1r * qr - g1i * qi, f1i = g1r * qi + g1i * qr;  // F_x(r1) W_N^{k2 r1}
            put(4, f0r - f1i, f0i + f1r);  // X_self       = F_x(r0) + i F_x(r1)
            put(5, f0r + f1i, f0i - f1r);  // X_twin input = F_x(r0) - i F_x(r1)
        } else if constexpr (KIND == GENF_PLTF) {
#pragma unroll
            for (int j = 0; j < 6; j++) putp(j, accr[j], acci[j]);  // X, Y, Z, fX, fY, fZ
        } else if constexpr (KIND == GENF_PLTN) {
            putp(0, -acci[0] - accr[3], accr[0] - acci[3]);   // JOB_XV_SELF (i - f) s_x D = i X - fX
            putp(1, -acci[0] + accr[3], accr[0] + acci[3]);   // JOB_XV_TWIN (i + f) s_x D = i X + fX
            putp(2, -accr[2] - acci[1], -acci[2] + accr[1]);  // JOB_B_SELF  -Z + i Y
            putp(3, accr[2] - acci[1], acci[2] + accr[1]);    // JOB_B_TWIN   Z + i Y
            putp(4, -accr[5] - acci[4], -acci[5] + accr[4]);  // JOB_D_SELF  -fZ + i fY
            putp(5, accr[5] - acci[4], acci[5] + accr[4]);    // JOB_D_TWIN   fZ + i fY
        } else {
            double xr, xi, yr, yi, zr, zi2;
            if constexpr (KIND == GENF_ZA) {
                xr = (double) kx * accr[1]; xi = (double) kx * acci[1];
                yr = (double) ky * accr[1]; yi = (double) ky * acci[1];
                zr = accr[2];               zi2 = acci[2];
            } else {
                xr = accr[1]; xi = acci[1];
                yr = accr[2]; yi = acci[2];
                zr = accr[3]; zi2 = acci[3];
            }
            putp(0, accr[0] - xr, acci[0] - xi);   // JOB_A_SELF  (1 - s_x) D
            putp(1, accr[0] + xr, acci[0] + xi);   // JOB_A_TWIN  (1 + s_x) D
            putp(2, -zr - yi, -zi2 + yr);          // JOB_B_SELF  (-s_z + i s_y) D
            putp(3, zr - yi, zi2 + yr);            // JOB_B_TWIN  ( s_z + i s_y) D
            if constexpr (KIND == GENF_PLT) {
                putp(4, -accr[4], -acci[4]);                        // JOB_C_BOTH  -f s_x D
                putp(5, -accr[6] - acci[5], -acci[6] + accr[5]);    // JOB_D_SELF  f(-s_z + i s_y) D
                putp(6, accr[6] - acci[5], acci[6] + accr[5]);      // JOB_D_TWIN  f( s_z + i s_y) D
            }
        }
    }
    return vsum;
}

// genf_tile for the PLT kinds with the kz mirror folded in (round 4).  The modes (kx, ky, kz) and (kx, ky, -kz) of one column share
// |k|^2 — hence the zero rule, P(k) and 1/k^2 — AND the eigenmode: the table holds the +kz half space, e(-kz) = (e_x, e_y, -e_z),
// lambda the same (src/zeldovich.cpp:239-247), so k.e, the normalisation, f and the rescale factor are the same numbers: more than
// half of a PLT mode's ~350 instructions.  Only the draws, Box-Muller and the sums stay separate.  In the folded index
// z = k2 + L k1 the mirror of (k2, k1) is (L - k2, R - 1 - k1): a thread owns ZR lines k2 of the LOWER half [0, L/2] and their
// mirror lines L - k2, with a second RNG walker that runs the mirrored order (every move of the first walker negated).  The lines
// k2 = 0 and k2 = L/2 are their own mirrors (z -> N - z stays on the line): they are walked unpaired.
//   tiles: (x block, kz_chunks(L, ZR) chunks of the lower half, row)
template <int ZR, int KIND, bool PLAW>
__device__ __forceinline__ double genf_tile_kz(const GenConst &g, const GenJumps &J, const StoreLayout &S, const double *T, int zW,
                                               int ky0, int kyl_arg, int kyl_first, int nky, int L, int residue, int bx, int by,
                                               const cplx *__restrict__ twN, cplx *__restrict__ Y) {
    static_assert(KIND == GENF_PLTN || KIND == GENF_PLTF, "PLT kinds of the packed / field stores");
    constexpr bool BLK = KIND == GENF_PLTF;
    double vsum = 0.0;
    const int N = g.N, half = g.half, R = N / L;
    const int rsub = BLK ? (int) (threadIdx.x & (FIELD_RB - 1)) : 0;
    const int x    = BLK ? bx * (GEN_BX / FIELD_RB) + (int) (threadIdx.x / FIELD_RB) : bx * GEN_BX + (int) threadIdx.x;
    const int kyl  = BLK ? kyl_arg * FIELD_RB + rsub : kyl_arg;
    const int k20 = by * ZR;
    const int ky  = ky0 + kyl * S.ky_stride;  // >= 1
    if (x >= N || kyl < kyl_first) return 0.0;
    const int kx = x > half ? x - N : x;
    if (S.prune & 1) {  // see genf_tile
        bool all_zero = true;
        const int xt0 = x - x % zW;
        if constexpr (BLK) {
            for (int i = 0; i < zW; i++)
                for (int r = 0; r < FIELD_RB; r++)
                    all_zero = all_zero && column_is_zero(S, (xt0 + i) > half ? xt0 + i - N : xt0 + i,
                                                          ky0 + (kyl_arg * FIELD_RB + r) * S.ky_stride);
        } else {
            for (int i = -1; i <= zW; i++) {
                const int xi = modn(N, xt0 + i + N);
                all_zero = all_zero && column_is_zero(S, xi > half ? xi - N : xi, ky);
            }
        }
        if (all_zero) return 0.0;
    }
    const int kxy2  = kx * kx + ky * ky;
    const bool dead = (kx < 0 ? -kx : kx) == g.kmax || ky == g.kmax;  // zeldovich.cpp:350
    EigXY exy = {};
    if (!g.eig_lines) exy = eig_xy(g, eig_axis(g, eig_index_x(g, kx)), eig_axis(g, ky));
    const uint64_t offx = (uint64_t) (kx & 65535);
    // walker A at (k2 = k20, k1 = 0): z = k20 <= L/2 <= N/2; walker B at the mirror of the first PAIRED line (k2 = max(k20, 1),
    // k1 = 0), z' = N - k2, kz' = -k2; both one step ahead of their mode's counter
    u128 sA = advance_bits(g.row_state[ky], 2ULL * ((uint64_t) (k20 & 65535) * 65536ULL + offx) + 1ULL);
    u128 sB = advance_bits(g.row_state[ky], 2ULL * ((uint64_t) ((-(k20 > 0 ? k20 : 1)) & 65535) * 65536ULL + offx) + 1ULL);
    const int nz = by == kz_chunks(L, ZR) - 1 ? L / 2 + 1 - k20 : ZR;  // the last chunk ends with the self-mirrored line k2 = L/2
#pragma unroll 1
    for (int zi = 0; zi < nz; zi++) {
        const int k2 = k20 + zi, k2m = L - k2;
        const bool pair = k2 != 0 && 2 * k2 != L;  // (tile-uniform)
        double aAr[6], aAi[6], aBr[6], aBi[6];
#pragma unroll
        for (int j = 0; j < 6; j++) aAr[j] = aAi[j] = aBr[j] = aBi[j] = 0.0;
        // the fold terms of the line; PAIR (compile time) = the mirror line rides along.  Two copies of the loop so that the paired
        // one has no branch between the two modes' Box-Muller chains: the scheduler interleaves them (two waves per SIMD only)
        auto fold = [&](auto pair_c) {
            constexpr bool PAIR = decltype(pair_c)::value;
        EigRaw enext = {};
        if (g.eig_lines) enext = eig_lines_load(g, k2 > half ? k2 - N : k2, kyl, x);  // the entries of the first fold term (k2 <= L/2)
#pragma unroll 1
        for (int k1 = 0; k1 < R; k1++) {
            const int z  = k2 + L * k1, zm = N - z;  // zm: the mirror position (paired lines: 0 < z < N, z != N/2)
            const int kz = z > half ? z - N : z;
            const EigRaw ecur = enext;
            if (g.eig_lines && k1 + 1 < R) {  // request the next fold term's entries now: they arrive while this mode is computed
                const int zq = z + L, kzq = zq > half ? zq - N : zq;
                // (only if that term has a live mode in this wave — the same test its own iteration makes: at k_cutoff = 2 seven
                // fold terms in eight are dead, and requesting their entries cost PPD=8192 PLT 1.7 s)
                const bool liveq = !dead && (kzq < 0 ? -kzq : kzq) != g.kmax && (g.corner_modes || kxy2 + kzq * kzq < g.k2i_cut);
                if (__any(liveq)) enext = eig_lines_load(g, kzq, kyl, x);
            }
            const int k2i = kxy2 + kz * kz;
            const bool live = !dead && (kz < 0 ? -kz : kz) != g.kmax && (g.corner_modes || k2i < g.k2i_cut);
            const bool last = k1 + 1 >= R;
            const int zn = last ? k2 + 1 : z + L, zmn = N - zn;  // next positions of the two walkers
            const int selA = (z > half) != (zn > half), selB = (zm > half) != (zmn > half);
            const bool any = __any(live);
            zdpcg::Affine mA, mB;
            if (!last) {
                mA = any ? J.fwd[selA] : J.fwd_full[selA];
                mB = any ? J.mfwd[selB] : J.mfwd_full[selB];
            } else {
                mA = any ? J.back[selA] : J.back_full[selA];
                mB = any ? J.mback[selB] : J.mback_full[selB];
            }
            if (!any) {  // all 64 columns zeroed at this |kz|: only the walks move on
                sA = zdpcg::apply(mA, sA);
                if constexpr (PAIR) sB = zdpcg::apply(mB, sB);
                continue;
            }
            const uint64_t r1A = zdpcg::output(sA);
            const u128 tA      = zdpcg::step(sA);
            const uint64_t r2A = zdpcg::output(tA);
            sA = zdpcg::apply(mA, tA);
            uint64_t r1B = 0, r2B = 0;
            if constexpr (PAIR) {
                r1B = zdpcg::output(sB);
                const u128 tB = zdpcg::step(sB);
                r2B = zdpcg::output(tB);
                sB  = zdpcg::apply(mB, tB);
            }
            // ---- shared by the two modes: P(k), 1/k^2, the eigenmode (its z component changes sign), f, rescale ----
            const double k2v = (double) k2i * g.fundamental2;
            const double P   = genf_power<PLAW>(g, T, k2v);
            const double ik2 = frcp(k2v);
            double e[4];
            if (g.eig_lines)  // (uniform) the slab's (x, y)-interpolated lines, requested one fold term ahead
                eig_lines_finish(g.fundamental, kx, ky, kz, ecur, e);
            else
                eigenmode_fast(g, kx, ky, kz, exy, eig_axis(g, eig_index_z(g, kz)), e);
            const double f = (sqrt_pos(1. + 24 * e[3] * g.f_cluster) - 1) * .25;
            double rescale = 1.0;
            if (g.qPLTrescale) rescale = fexp(g.ln_growth_ratio * (g.target_f - f), T);
            const double sx = rescale * e[0], sy = rescale * e[1], sz = rescale * e[2];  // e = the coefficients s_j (eig_coeff_tail)
            auto one = [&](uint64_t r1, uint64_t r2, int kf, double szs, double (&ar)[6], double (&ai)[6]) {
                // cgauss<2> (power_spectrum.cpp:338-359); zeroed lanes of a live wave ride along with amplitude 0
                const uint64_t m1 = r1 + 1ULL;
                double v = P;
                if (!g.fixed_power) v = -P * flog(u64_to_double(m1), 64, T);
                v = (m1 == 0 && !g.fixed_power) || !live ? 0.0 : v;
                const double amp = sqrt_pos(v);
                double sn, cs;
                sincos_u01(u64_to_double(r2 + 1ULL), T, sn, cs);
                double dr = amp * cs, di = amp * sn;
                vsum = fma(dr, dr, fma(di, di, vsum));
                if (R > 1) {  // W_R^{kf r}: kf = the mode's own fold index
                    const cplx w = twN[modn(N, kf * residue * L)];
                    const double a = dr * w.x - di * w.y, b = dr * w.y + di * w.x;
                    dr = a;
                    di = b;
                }
                cmac(ar[0], ai[0], sx, dr, di);
                cmac(ar[1], ai[1], sy, dr, di);
                cmac(ar[2], ai[2], szs, dr, di);
                cmac(ar[3], ai[3], f * sx, dr, di);
                cmac(ar[4], ai[4], f * sy, dr, di);
                cmac(ar[5], ai[5], f * szs, dr, di);
            };
            one(r1A, r2A, k1, sz, aAr, aAi);
            if constexpr (PAIR) one(r1B, r2B, R - 1 - k1, -sz, aBr, aBi);
        }
        };
        if (pair)
            fold(std::true_type{});
        else
            fold(std::false_type{});
        // ---- outputs of line k2 (and of its mirror line k2m) from the six sums, times W_N^{line r} ----
        auto emit = [&](int line, const double (&ar)[6], const double (&ai)[6]) {
            double pr = 1.0, pi = 0.0;
            if (R > 1) {
                const cplx w = twN[modn(N, line * residue)];
                pr = w.x;
                pi = w.y;
            }
            auto putp = [&](int j, double vr, double vi) {
                const unsigned idx = BLK ? y_index_blocked(j, kyl, nky, L, line, N, x) : (unsigned) (((j * nky + kyl) * L + line) * N + x);
                Y[idx] = cplx{vr * pr - vi * pi, vr * pi + vi * pr};
            };
            if constexpr (KIND == GENF_PLTF) {
#pragma unroll
                for (int j = 0; j < 6; j++) putp(j, ar[j], ai[j]);  // X, Y, Z, fX, fY, fZ
            } else {
                putp(0, -ai[0] - ar[3], ar[0] - ai[3]);   // JOB_XV_SELF (i - f) s_x D = i X - fX
                putp(1, -ai[0] + ar[3], ar[0] + ai[3]);   // JOB_XV_TWIN (i + f) s_x D = i X + fX
                putp(2, -ar[2] - ai[1], -ai[2] + ar[1]);  // JOB_B_SELF  -Z + i Y
                putp(3, ar[2] - ai[1], ai[2] + ar[1]);    // JOB_B_TWIN   Z + i Y
                putp(4, -ar[5] - ai[4], -ai[5] + ar[4]);  // JOB_D_SELF  -fZ + i fY
                putp(5, ar[5] - ai[4], ai[5] + ar[4]);    // JOB_D_TWIN   fZ + i fY
            }
        };
        emit(k2, aAr, aAi);
        if (pair) emit(k2m, aBr, aBi);
    }
    return vsum;
}

// genf_tile for the ZA kinds with the x mirror folded in: a thread owns the columns kx = +xh and kx = -xh (x = xh
// and N - xh).  The two modes of a (ky, kz) share |k|^2, hence the zero rule, P(k) and 1/k^2 — a quarter of the
// arithmetic of a mode; the two RNG walks, Box-Muller draws and field sums stay separate.  (The PLT kinds keep
// genf_tile: their eigenvectors differ between +kx and -kx and their field sums already fill the registers.)
template <int ZR, int KIND, bool PLAW>
__device__ __forceinline__ double genf_tile_mirror(const GenConst &g, const GenJumps &J, const StoreLayout &S, const double *T,
                                                   int zW, int ky0, int kyl_arg, int kyl_first, int nky, int L, int residue,
                                                   int residue2, int bx, int by, const cplx *__restrict__ twN,
                                                   cplx *__restrict__ Y) {
    static_assert(KIND == GENF_DENS || KIND == GENF_ZA || KIND == GENF_ZAP || KIND == GENF_ZAF || KIND == GENF_ZAFD, "ZA kinds only");
    constexpr bool ZA2 = KIND == GENF_ZAP || KIND == GENF_ZAF || KIND == GENF_ZAFD;  // two residues per pass
    constexpr int NACC = KIND == GENF_DENS ? 1 : KIND == GENF_ZA ? 3 : KIND == GENF_ZAFD ? 6 : 4;
    double vsum = 0.0;
    const int N = g.N, half = g.half, R = N / L;
    // Field store (GENF_ZAF): the folded inputs and the store are laid out in blocks of FIELD_RB = 8 rows x 1 column (one
    // 128-byte line, zd_device.h), so a workgroup takes 8 rows x 32 columns (kyl_arg = the row GROUP) and a lane's 7
    // neighbours hold the other rows of its column; elsewhere a workgroup is one row x GEN_BX columns.
    constexpr bool BLK = KIND == GENF_ZAF || KIND == GENF_ZAFD;
    const int rsub = BLK ? (int) (threadIdx.x & (FIELD_RB - 1)) : 0;
    const int xh   = BLK ? bx * (GEN_BX / FIELD_RB) + (int) (threadIdx.x / FIELD_RB) : bx * GEN_BX + (int) threadIdx.x;  // 0 .. N/2
    const int kyl  = BLK ? kyl_arg * FIELD_RB + rsub : kyl_arg;
    const int k20 = by * ZR;
    const int ky  = ky0 + kyl * S.ky_stride;  // >= 1
    if (xh > half || kyl < kyl_first) return 0.0;  // (the row ky = 0 of a first group belongs to k_gen)
    const bool hasB = xh > 0 && xh < half;  // x = 0 and x = N/2 are their own mirrors
    const int xA = xh, xB = hasB ? N - xh : xh;
    auto tile_zero = [&](int x) {  // see k_gen: the z-FFT tile(s) this column belongs to are all zero
        bool all_zero = true;
        if constexpr (BLK) {  // k_zfft_fb tile: zW columns x the 8 rows of the group
            const int xt0 = x - x % zW;
            for (int i = 0; i < zW; i++)
                for (int r = 0; r < FIELD_RB; r++)
                    all_zero = all_zero && column_is_zero(S, (xt0 + i) > half ? xt0 + i - N : xt0 + i,
                                                          ky0 + (kyl_arg * FIELD_RB + r) * S.ky_stride);
            return all_zero;
        }
        const int xt0 = x - x % zW;
        for (int i = -1; i <= zW; i++) {
            const int xi = modn(N, xt0 + i + N);
            all_zero = all_zero && column_is_zero(S, xi > half ? xi - N : xi, ky);
        }
        return all_zero;
    };
    bool needA = true, needB = hasB;
    if (S.prune & 1) {
        needA = !tile_zero(xA);
        needB = hasB && !tile_zero(xB);
        if (!needA && !needB) return 0.0;
    }
    const int kxy2  = xh * xh + ky * ky;
    const bool dead = xh == g.kmax || ky == g.kmax;  // zeldovich.cpp:350
    u128 sA, sB;
    {  // states one step ahead of the first modes' counters
        const int kz0 = k20 > half ? k20 - N : k20;  // k20 > N/2 only happens when R = 1
        const uint64_t offz = (uint64_t) (kz0 & 65535) * 65536ULL;
        sA = advance_bits(g.row_state[ky], 2ULL * (offz + (uint64_t) (xh & 65535)) + 1ULL);
        sB = advance_bits(g.row_state[ky], 2ULL * (offz + (uint64_t) ((hasB ? -xh : xh) & 65535)) + 1ULL);
    }
    const zdpcg::Affine fwd0 = J.fwd[0], fwdf0 = J.fwd_full[0];
#pragma unroll 1
    for (int zi = 0; zi < ZR; zi++) {
        const int k2 = k20 + zi;
        double aAr[NACC], aAi[NACC], aBr[NACC], aBi[NACC];
#pragma unroll
        for (int j = 0; j < NACC; j++) aAr[j] = aAi[j] = aBr[j] = aBi[j] = 0.0;
        // One fold term k1 of this k2.  ZA2 kinds (two residues r and r + R/2 per pass): the second residue's fold twiddle is
        // (-1)^k1 times the first one's, so E(r0) = Ee + Eo, E(r1) = Ee - Eo with Ee / Eo the sums over the even / odd terms
        // (likewise Z): a mode updates ONE pair of sums (4 FMAs instead of 8); the loop is unrolled by two so that the parity
        // is a compile-time index, and the sums are combined after it.  Slots: [0] Ee, [1] Ze, [2] Eo, [3] Zo.
        constexpr int NPAR = ZA2 ? 2 : 1;  // (ZA2: R is a power of two >= 2)
#pragma unroll 1
        for (int k10 = 0; k10 < R; k10 += NPAR) {
#pragma unroll
        for (int PAR = 0; PAR < NPAR; PAR++) {
            const int k1 = k10 + PAR;
            const int z  = k2 + L * k1;
            const int kz = z > half ? z - N : z;
            const int k2i = kxy2 + kz * kz;
            const bool live = !dead && (kz < 0 ? -kz : kz) != g.kmax && (g.corner_modes || k2i < g.k2i_cut);
            const bool last = k1 + 1 >= R;
            const int sel   = (z > half) != ((last ? k2 + 1 : z + L) > half);
            const bool any  = __any(live);
            zdpcg::Affine m;
            if (!last && !sel)
                m = any ? fwd0 : fwdf0;
            else if (!last)
                m = any ? J.fwd[1] : J.fwd_full[1];
            else
                m = any ? J.back[sel] : J.back_full[sel];
            if (!any) {
                sA = zdpcg::apply(m, sA);
                sB = zdpcg::apply(m, sB);
                continue;
            }
            const uint64_t r1A = zdpcg::output(sA), r1B = zdpcg::output(sB);
            const u128 tA = zdpcg::step(sA), tB = zdpcg::step(sB);
            const uint64_t r2A = zdpcg::output(tA), r2B = zdpcg::output(tB);
            sA = zdpcg::apply(m, tA);
            sB = zdpcg::apply(m, tB);
            // ---- shared by the two modes: P(k), 1/k^2 ----
            const double k2v = (double) k2i * g.fundamental2;
            const double P   = genf_power<PLAW>(g, T, k2v);
            const double q   = g.fundamental * frcp(k2v);
            // W_R^{k1 r}
            double wr = 1.0, wi = 0.0;
            if (R > 1) {
                const cplx w = twN[modn(N, k1 * residue * L)];
                wr = w.x;
                wi = w.y;
            }
            const double dkz = (double) kz;
            const double qwr = q * wr, qwi = q * wi;  // ZA2: fundamental / k^2 folded into the twiddle (shared by the two modes)
            auto one = [&](uint64_t r1, uint64_t r2, double (&ar)[NACC], double (&ai)[NACC]) {
                // cgauss<2> (power_spectrum.cpp:338-359); zeroed lanes of a live wave ride along with amplitude 0
                const uint64_t m1 = r1 + 1ULL;
                double v = P;
                if (!g.fixed_power) v = -P * flog(u64_to_double(m1), 64, T);
                v = (m1 == 0 && !g.fixed_power) || !live ? 0.0 : v;
                const double amp = sqrt_pos(v);
                double sn, cs;
                sincos_u01(u64_to_double(r2 + 1ULL), T, sn, cs);
                const double d0r = amp * cs, d0i = amp * sn;
                if (g.accum_var) vsum += v;  // |D|^2 = amp^2 (cos^2 + sin^2); wanted by the first pass of a run only
                if constexpr (ZA2) {
                    const double er = d0r * qwr - d0i * qwi, ei = d0r * qwi + d0i * qwr;
                    ar[2 * PAR] += er;
                    ai[2 * PAR] += ei;
                    cmac(ar[2 * PAR + 1], ai[2 * PAR + 1], dkz, er, ei);
                    if constexpr (KIND == GENF_ZAFD) {  // the density sum: D times the fold twiddle alone; slots [4] De, [5] Do
                        ar[4 + PAR] += d0r * wr - d0i * wi;
                        ai[4 + PAR] += d0r * wi + d0i * wr;
                    }
                } else {
                    const double dr = d0r * wr - d0i * wi, di = d0r * wi + d0i * wr;
                    ar[0] += dr;
                    ai[0] += di;
                    if constexpr (KIND == GENF_ZA) {
                        const double er = q * dr, ei = q * di;
                        ar[1] += er;
                        ai[1] += ei;
                        cmac(ar[2], ai[2], dkz, er, ei);
                    }
                }
            };
            one(r1A, r2A, aAr, aAi);
            one(r1B, r2B, aBr, aBi);
        }
        }
        if constexpr (ZA2) {
            auto combine = [](double (&a)[NACC]) {
                const double ee = a[0], ze = a[1], eo = a[2], zo = a[3];
                a[0] = ee + eo;  // E(r0)
                a[1] = ze + zo;  // Z(r0)
                a[2] = ee - eo;  // E(r1)
                a[3] = ze - zo;  // Z(r1)
                if constexpr (KIND == GENF_ZAFD) {
                    const double de = a[4], dd = a[5];
                    a[4] = de + dd;  // D(r0)
                    a[5] = de - dd;  // D(r1)
                }
            };
            combine(aAr);
            combine(aAi);
            combine(aBr);
            combine(aBi);
        }
        // ---- job inputs from the field sums (genf_tile), for column xA with kx = +xh and column xB with kx = -xh ----
        double pr = 1.0, pi = 0.0, qr = 1.0, qi = 0.0;
        if (R > 1) {
            const cplx w = twN[modn(N, k2 * residue)];
            pr = w.x;
            pi = w.y;
            if constexpr (ZA2) {
                const cplx w2 = twN[modn(N, k2 * residue2)];
                qr = w2.x;
                qi = w2.y;
            }
        }
        auto emit = [&](int x, double dkx, const double (&ar)[NACC], const double (&ai)[NACC]) {
            auto put = [&](int j, double vr, double vi) {
                const unsigned idx = BLK ? y_index_blocked(j, kyl, nky, L, k2, N, x) : (unsigned) (((j * nky + kyl) * L + k2) * N + x);
                Y[idx] = cplx{vr, vi};
            };
            auto putp = [&](int j, double vr, double vi) { put(j, vr * pr - vi * pi, vr * pi + vi * pr); };
            const double dky = (double) ky;
            if constexpr (KIND == GENF_DENS) {
                putp(0, ar[0], ai[0]);
            } else if constexpr (KIND == GENF_ZA) {
                const double xr = dkx * ar[1], xi = dkx * ai[1], yr = dky * ar[1], yi = dky * ai[1];
                putp(0, ar[0] - xr, ai[0] - xi);       // JOB_A_SELF
                putp(1, ar[0] + xr, ai[0] + xi);       // JOB_A_TWIN
                putp(2, -ar[2] - yi, -ai[2] + yr);     // JOB_B_SELF
                putp(3, ar[2] - yi, ai[2] + yr);       // JOB_B_TWIN
            } else if constexpr (KIND == GENF_ZAF || KIND == GENF_ZAFD) {
                auto putq = [&](int j, double vr, double vi) { put(j, vr * qr - vi * qi, vr * qi + vi * qr); };
                putp(0, ar[0], ai[0]);                 // E(r0)
                putp(1, ar[1], ai[1]);                 // Z(r0)
                putq(2, ar[2], ai[2]);                 // E(r1)
                putq(3, ar[3], ai[3]);                 // Z(r1)
                if constexpr (KIND == GENF_ZAFD) {
                    putp(4, ar[4], ai[4]);             // D(r0)
                    putq(5, ar[5], ai[5]);             // D(r1)
                }
            } else {
                auto putq = [&](int j, double vr, double vi) { put(j, vr * qr - vi * qi, vr * qi + vi * qr); };
                const double y0r = dky * ar[0], y0i = dky * ai[0], y1r = dky * ar[2], y1i = dky * ai[2];
                putp(0, -ar[1] - y0i, -ai[1] + y0r);   // JOB_B_SELF (r0)
                putp(1, ar[1] - y0i, ai[1] + y0r);     // JOB_B_TWIN (r0)
                putq(2, -ar[3] - y1i, -ai[3] + y1r);   // JOB_B_SELF (r1)
                putq(3, ar[3] - y1i, ai[3] + y1r);     // JOB_B_TWIN (r1)
                const double g0r = -dkx * ai[0], g0i = dkx * ar[0], g1r = -dkx * ai[2], g1i = dkx * ar[2];
                const double f0r = g0r * pr - g0i * pi, f0i = g0r * pi + g0i * pr;
                const double f1r = g1r * qr - g1i * qi, f1i = g1r * qi + g1i * qr;
                put(4, f0r - f1i, f0i + f1r);          // X_self       = F_x(r0) + i F_x(r1)
                put(5, f0r + f1i, f0i - f1r);          // X_twin input = F_x(r0) - i F_x(r1)
            }
        };
        if (needA) emit(xA, (double) xh, aAr, aAi);
        if (needB) emit(xB, (double) -xh, aBr, aBi);
    }
    // a thread without a mirror column (x = 0, N/2) ran its second walk on a copy of the first: count its modes once
    return hasB ? vsum : 0.5 * vsum;
}

// Persistent launch: `gridDim.x` workgroups pull tiles (x block, k2 chunk, row) from an atomic counter.  The grid
// is sized to a few workgroups per CU (zd_plan: gen_wgs_per_cu) so that the HBM-bound k_zfft of the previous
// slab, running on the second stream, always finds registers and LDS next to the generator's waves.
template <int ZR, int KIND, bool PLAW, bool MIRROR>
// (PLT kinds at __launch_bounds__(GEN_BX, 4) = 128 registers, 7-16 spilled dwords, two workgroups per CU so that a 512-thread
// z-FFT workgroup shares the CU: measured PPD=2048 PLT+rescale 558 -> 574 ms, PPD=4096 PLT 9.2 -> 10.7 s.  The PLT Z stage at
// PPD=2048 is HBM-bound — generator writes 0.41 TB of folded inputs, the z FFT reads them and writes 0.41 TB of store: 1.23 TB
// in 0.26 s — so overlapping the two kernels buys nothing there, and at PPD=4096 the capped generator is simply slower.)
// The ZA kinds are held to 128 registers (three generator workgroups + one 256-thread z-FFT workgroup of 128 registers fill a
// SIMD's 512 exactly; at 129 the allocation granule of 8 makes it 136 and the z FFT no longer fits beside them).
// (GENF_ZAFD — six fields, ZD_qdensity on the composite grids — is left at three workgroups per CU: its two extra pairs of sums do not fit 128)
// (PLT kinds, MIRROR = the kz-paired form genf_tile_kz: 201-209 registers, two workgroups per CU + the z FFT beside them.  Held to 168
// — three workgroups per CU, 46 spilled dwords — the generator alone is faster, 4.36 s against 4.72 s per step at PPD=4096 PLT, but
// then no z-FFT workgroup fits a CU beside three of them and the Z stage takes 5.45 s against 4.97 s: profiles/r04_tuning_notes.md)
__global__ __launch_bounds__(GEN_BX, (KIND == GENF_ZAF || KIND == GENF_ZAP || KIND == GENF_ZA || KIND == GENF_DENS) ? 4 : (KIND == GENF_ZAFD ? 3 : 1))
void k_genf(GenConst g, GenJumps J, StoreLayout S, int zW, int ky0, int kyl0, int nky,
                                                 int nrows, int L, int residue, int residue2,
                                                 const cplx *__restrict__ twN, cplx *__restrict__ Y,
                                                 unsigned *__restrict__ tile_ctr) {
    extern __shared__ __attribute__((aligned(16))) double T[];  // GenfTab image (+ one slot for the tile index)
    for (int i = threadIdx.x; i < g.genf_n / 2; i += GEN_BX)
        reinterpret_cast<double2 *>(T)[i] = reinterpret_cast<const double2 *>(g.genf_tab)[i];
    unsigned *slot = reinterpret_cast<unsigned *>(T + g.genf_n);
    // PLT kinds of the packed / field stores: MIRROR selects the kz-paired form (genf_tile_kz), a kernel of its own so that the
    // unpaired one keeps its 148 registers (three workgroups per CU)
    constexpr bool KZ = MIRROR && (KIND == GENF_PLTN || KIND == GENF_PLTF);
    constexpr bool XMIR = MIRROR && !KZ;  // ZA kinds: the x mirror (genf_tile_mirror)
    constexpr bool BLK = (XMIR && (KIND == GENF_ZAF || KIND == GENF_ZAFD)) || KIND == GENF_PLTF;  // 8 rows x 32 columns per workgroup
    constexpr int XW = BLK ? GEN_BX / FIELD_RB : GEN_BX;
    const int gx = ((XMIR ? g.N / 2 + 1 : g.N) + XW - 1) / XW, gy = KZ ? kz_chunks(L, ZR) : L / ZR;
    const int gz = BLK ? nky / FIELD_RB : nrows;  // row groups of the whole slab (lanes of rows < kyl0 idle), or rows
    const unsigned ntiles = (unsigned) (gx * gy * gz);
    double vsum = 0.0;
    for (;;) {
        __syncthreads();  // table image complete / previous tile index consumed
        if (threadIdx.x == 0) *slot = atomicAdd(tile_ctr, 1u);
        __syncthreads();
        const unsigned tile = *slot;
        if (tile >= ntiles) break;
        const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
        if constexpr (KZ)
            vsum += genf_tile_kz<ZR, KIND, PLAW>(g, J, S, T, zW, ky0, BLK ? bz : kyl0 + bz, kyl0, nky, L, residue, bx, by, twN, Y);
        else if constexpr (MIRROR)
            vsum += genf_tile_mirror<ZR, KIND, PLAW>(g, J, S, T, zW, ky0, BLK ? bz : kyl0 + bz, kyl0, nky, L, residue, residue2, bx, by,
                                                     twN, Y);
        else
            vsum += genf_tile<ZR, KIND, PLAW>(g, J, S, T, zW, ky0, BLK ? bz : kyl0 + bz, kyl0, nky, L, residue, residue2, bx, by, twN, Y);
    }
    if (g.accum_var) {  // every lane is back here: wave sum, one atomic per wave; rows ky >= 1 stand for their twins too
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&g.var_slots[(blockIdx.x * (GEN_BX / 64) + (threadIdx.x >> 6)) % NSLOT], 2.0 * vsum);
    }
}

#ifdef ZD_TESTING
// test hook: raw draws / amplitudes for an explicit mode list (counter addressing, no walk)
__global__ void k_test_modes(GenConst g, long long n, const int *__restrict__ kxyz, uint64_t *__restrict__ draws,
                             double *__restrict__ D) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int kx = kxyz[3 * i], ky = kxyz[3 * i + 1], kz = kxyz[3 * i + 2];
    const uint64_t off = 2ULL * ((uint64_t) (kz & 65535) * 65536ULL + (uint64_t) (kx & 65535));
    u128 s = advance_bits(g.row_state[ky], off);
    s = zdpcg::step(s);
    const uint64_t r1 = zdpcg::output(s);
    s = zdpcg::step(s);
    const uint64_t r2 = zdpcg::output(s);
    if (draws) {
        draws[2 * i]     = r1;
        draws[2 * i + 1] = r2;
    }
    if (D) {
        const double k2 = (double) (kx * kx + ky * ky + kz * kz) * g.fundamental2;
        double dr = 0, di = 0;
        if (!mode_is_zero(g, kx, ky, kz, k2)) {
            if (g.is_powerlaw)
                gauss_mode<true>(g, k2, r1, r2, dr, di);
            else
                gauss_mode<false>(g, k2, r1, r2, dr, di);
        }
        D[2 * i]     = dr;
        D[2 * i + 1] = di;
    }
}

// test hook: the PRODUCTION arithmetic of k_genf (LDS-table ln / exp / sincos / spline, integer zero rule, frcp) on an
// explicit mode list; out[3*i] = {Re D, Im D, fundamental / k^2}
template <bool PLAW>
__global__ void k_test_modes_table(GenConst g, long long n, const int *__restrict__ kxyz, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double T[];
    for (int i = threadIdx.x; i < g.genf_n / 2; i += blockDim.x)
        reinterpret_cast<double2 *>(T)[i] = reinterpret_cast<const double2 *>(g.genf_tab)[i];
    __syncthreads();
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int kx = kxyz[3 * i], ky = kxyz[3 * i + 1], kz = kxyz[3 * i + 2];
    const uint64_t off = 2ULL * ((uint64_t) (kz & 65535) * 65536ULL + (uint64_t) (kx & 65535)) + 1ULL;
    const u128 s      = advance_bits(g.row_state[ky], off);
    const uint64_t r1 = zdpcg::output(s);
    const uint64_t r2 = zdpcg::output(zdpcg::step(s));
    const int k2i     = kx * kx + ky * ky + kz * kz;
    const bool dead   = (kx < 0 ? -kx : kx) == g.kmax || ky == g.kmax || (kz < 0 ? -kz : kz) == g.kmax;
    const bool live   = !dead && (g.corner_modes || k2i < g.k2i_cut) && k2i > 0;
    const double k2v  = (double) (k2i > 0 ? k2i : 1) * g.fundamental2;
    const double P    = genf_power<PLAW>(g, T, k2v);
    const double ik2  = frcp(k2v);
    const uint64_t m1 = r1 + 1ULL;
    double v = P;
    if (!g.fixed_power) v = -P * flog(u64_to_double(m1), 64, T);
    v = (m1 == 0 && !g.fixed_power) || !live ? 0.0 : v;
    const double amp = sqrt_pos(v);
    double sn, cs;
    sincos_u01(u64_to_double(r2 + 1ULL), T, sn, cs);
    out[3 * i]     = amp * cs;
    out[3 * i + 1] = amp * sn;
    out[3 * i + 2] = g.fundamental * ik2;
}

#endif  // ZD_TESTING
// ------------------------------------------------------------------------------------------------
// k_zfft: length-L FFT of one job's folded inputs for a W-wide column tile of row ky, then the
// Hermitian stores: "self" columns at (row ky, column x), "twin" columns conjugated at
// (row N-ky, column N-x) — same plane index, see DESIGN.md §2.2.  Replaces InverseFFT_Yonly +
// StoreBlock (zeldovich.cpp:508-511, block_array.cpp:387-414).
//   grid: (N/W, nky, njobs)   block: W*L/E
template <int L, int E, int W>
__global__ __launch_bounds__(W *L / E) void k_zfft(JobList jobs, StoreLayout S, int ky0, int kyloc0, int nky, int Zq,
                                                  const cplx *__restrict__ Y, const cplx *__restrict__ twL,
                                                  cplx *__restrict__ out) {
    using PL  = zdfft::Plan<L, E>;
    using LDS = zdfft::ColsInner<L, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int N = S.N;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    const int kyl = blockIdx.y;
    const int ky  = ky0 + kyl * S.ky_stride;
    const int kind = jobs.kind[blockIdx.z];
    const bool twin_only = jobs.twin[blockIdx.z] != 0;
    // twin columns are stored mirrored (column N-x): shifting the tile of twin-only jobs by one column
    // makes the mirrored run start on a tile boundary, i.e. whole 128-byte lines instead of 112 + 16 B
    const int x = (blockIdx.x * W + w + (twin_only ? 1 : 0)) & (N - 1);
    if (ky == 0 && twin_only) return;  // ky = 0 is its own twin plane: every column written as "self"
    const int lZq = 31 - __clz(Zq);
    if (S.prune & 2) {  // tile of identically-zero columns: k_gen produced nothing, k_yfft will not read it
        if (__syncthreads_and(column_is_zero(S, x > S.half ? x - N : x, ky))) return;
    }
    // one 64-bit scalar base per workgroup + 32-bit byte offsets (a job's column tile spans L*N*16 B < 4 GB)
    const char *src = reinterpret_cast<const char *>(Y + (((long long) blockIdx.z * nky + kyl) * L) * N);
    const unsigned xb = (unsigned) x * 16u, rb = (unsigned) N * 16u;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        cplx v = cplx{1.0 + e, 2.0 * t};
        if (!ZD_TUNE(S.prune & 8)) v = ld_stream(reinterpret_cast<const cplx *>(src + ((unsigned) (t + T * e) * rb + xb)), ZD_TUNE(S.nt & 16));  // bit 3: ablation
        re[e] = v.x;
        im[e] = v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, twL);
    if (ZD_TUNE(S.prune & 16) && re[0] != 123.456) return;  // bit 4: tuning ablation (skip stores)

    const int arr = jobs.arr[blockIdx.z];
    const bool st_self = !twin_only;
    const bool st_twin = (ky != 0) && (twin_only || kind == JOB_C_BOTH || kind == JOB_DENS);
    const double sgr = (kind == JOB_C_BOTH) ? -1.0 : 1.0, sgi = (kind == JOB_C_BOTH) ? 1.0 : -1.0;  // -conj / conj
    const int loc_self = kyloc0 + kyl, loc_twin = S.Hq + kyloc0 + kyl;
    // the twin row sits a constant number of rows away from the self row (same plane, same array)
    const int drow = store_row(S, 0, 0, 0, loc_twin) - store_row(S, 0, 0, 0, loc_self);
    const int xt   = (N - x) & (N - 1);
    int t2 = t;
    asm volatile("" : "+v"(t2));  // keep the store-address arithmetic after the FFT (register pressure)
    // (plane-interleaved rows, StoreLayout::lq — the fused PLT Z stage's store, whose ky = 0 row comes through here: planes of a
    // group share a row, column x of plane zl sits at (x << lq) + zl % 2^lq; lq = 0 everywhere else)
    const int lq = S.lq, qm = (1 << lq) - 1;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int z2  = t2 + T * e;
        const int dst = z2 >> lZq, zl = z2 & (Zq - 1);  // Zq = 2^lZq
        const int row = store_row(S, dst, zl >> lq, arr, loc_self);
        if (st_self) st_stream(out + ((long long) row * S.pitch + ((long long) x << lq) + (zl & qm)), cplx{re[e], im[e]}, ZD_TUNE(S.nt & 8));
        if (st_twin)
            st_stream(out + ((long long) (row + drow) * S.pitch + ((long long) xt << lq) + (zl & qm)), cplx{sgr * re[e], sgi * im[e]}, ZD_TUNE(S.nt & 8));
    }
}

// ------------------------------------------------------------------------------------------------
// k_yfft: in-place y FFT of the block store.  grid: (N/W, narray, nplanes)  block: W*N/E
template <int N, int E, int W, int MINW = 1, bool ONEBLOCK = false>
__global__ __launch_bounds__(W *N / E, MINW) void k_yfft(StoreLayout S, const cplx *__restrict__ tw, cplx *__restrict__ data) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    // Tiles narrower than a 128-byte line (W < 8) share their lines with G - 1 neighbours.  Workgroups go to the
    // XCDs round-robin by blockIdx.x, so neighbours would sit behind different L2s and every line would be fetched
    // (and written back in pieces) G times: give the G tiles of a line to workgroups b, b + 8, ... of ONE XCD.
    constexpr int G = W >= 8 ? 1 : 8 / W;
    int tile = blockIdx.x;
    if constexpr (G > 1 && (N / W) % (8 * G) == 0) tile = ((tile / (8 * G)) * 8 + (tile % 8)) * G + ((tile / 8) % G);
    // plane-interleaved rows (StoreLayout::lq, single-rank stores only): the tile's W lines are W / 2^lq columns x 2^lq planes of the
    // plane group blockIdx.z — the same contiguous W x 16 bytes of every row; xq = position inside the interleaved row
    const int xq = tile * W + w, x = xq >> S.lq;
    const int zl = blockIdx.z, a = blockIdx.y;
    const int kxs = x > N / 2 ? x - N : x;
    // a tile none of whose columns has a live row: the z stage wrote nothing there and the x stage takes zeros for those
    // columns (EpiConst::xdead_lo / hi) — nothing to transform
    // (PRUNE_YTILE: set only together with xdead_lo / hi — the f_NL phi round's consumers k_xphi / k_yfwd / k_zfwd read every column,
    // so its plan keeps the per-element zero rule below but transforms every tile)
    if ((S.prune & PRUNE_YTILE) && __syncthreads_and(column_is_zero(S, kxs, 0))) return;
    double re[E], im[E];
    if constexpr (ONEBLOCK) {
        // single rank, all 2*Hq row slots of a (plane, array) contiguous: one 64-bit scalar base per
        // workgroup and 32-bit byte offsets per element (half the address registers, no spills)
        char *base = reinterpret_cast<char *>(data + (long long) store_row(S, 0, zl, a, 0) * S.pitch);
        const unsigned xb = (unsigned) xq * 16u, pb = (unsigned) S.pitch * 16u;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int ky = t + T * e;
            const bool skip = (ky == N / 2) || ((S.prune & 4) && column_is_zero(S, kxs, ky > N / 2 ? ky - N : ky));
            const unsigned slot = ky < N / 2 ? ky : N / 2 + (N - ky);
            cplx v = cplx{0.0, 0.0};
            if (!skip) v = ld_stream(reinterpret_cast<const cplx *>(base + (slot * pb + xb)), ZD_TUNE(S.nt & 1));
            re[e] = v.x;
            im[e] = v.y;
        }
        if (!ZD_TUNE(S.prune & 128)) zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);  // bit 7: tuning ablation
        if (ZD_TUNE(S.prune & 256) && re[0] != 123.456) return;                          // bit 8: tuning ablation (no stores)
        // the store offsets equal the load offsets; recompute them from a laundered thread index so the
        // compiler does not keep 16 offset registers alive (and spill them) across the whole FFT
        int t2 = t;
        asm volatile("" : "+v"(t2));
        // plane-interleaved rows (lq = 2, W = 8: the tile is one 128-byte line per row, 2 columns x 4 planes): the line goes back
        // with the two columns of a plane side by side — [plane][column] instead of [column][plane] — so that the x stage, which
        // transforms the lines of a plane PAIR, reads 64 contiguous bytes per column pair (whole sectors) instead of 32 of every 64
        unsigned xs = xb;
        if (S.lq == 2 && W % 8 == 0) xs = (unsigned) ((xq & ~7) + ((w & 3) << 1) + ((w >> 2) & 1)) * 16u;  // (W = 16: two such lines per row)
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int y = t2 + T * e;
            const unsigned slot = y < N / 2 ? y : (y == N / 2 ? N / 2 : N / 2 + (N - y));
            st_stream(reinterpret_cast<cplx *>(base + (slot * pb + xs)), cplx{re[e], im[e]}, ZD_TUNE(S.nt & 2));
        }
    } else {
        cplx *base = data + x;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int ky = t + T * e;
            // Nyquist row (zeldovich.cpp:644-650) and pruned columns are identically zero: not read
            const bool skip = (ky == N / 2) || ((S.prune & 4) && column_is_zero(S, kxs, ky > N / 2 ? ky - N : ky));
            cplx v = cplx{0.0, 0.0};
            if (!skip) v = base[row_offset(S, zl, a, ky)];
            re[e] = v.x;
            im[e] = v.y;
        }
        zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);
        int t2 = t;
        asm volatile("" : "+v"(t2));  // see above: do not carry the load offsets across the FFT
#pragma unroll
        for (int e = 0; e < E; e++) base[row_offset(S, zl, a, t2 + T * e)] = cplx{re[e], im[e]};
    }
}

// ------------------------------------------------------------------------------------------------
// Field store (PACK_ZAFIELD, zd_device.h): z stage and y stage.
//
// k_zfft_f: length-L FFT of one potential (E or Z of one residue) for NC columns x the FIELD_RB = 8 rows of a row group,
// read from Y[field][group][k2][x][8 rows] and stored at (row block, compact column, row in block): both sides move whole
// 128-byte lines (one column x 8 rows).  No Hermitian twins are written: the y stage rebuilds them from the symmetry
//      E(-ky, -kx) = conj E(ky, kx),   Z(-ky, -kx) = -conj Z(ky, kx)      (E, i Z: z-transforms of real fields)
//   grid: (N/NC, row groups, 4)   block: NC*8*L/E
template <int L, int E, int NC>
__global__ __launch_bounds__(NC *FIELD_RB *L / E) void k_zfft_f(FieldLayout F, StoreLayout S, int ky0, int kyloc0, int nky,
                                                              const cplx *__restrict__ Y, const cplx *__restrict__ twL,
                                                              cplx *__restrict__ out) {
    static_assert(NC <= FIELD_CW && FIELD_CW % NC == 0, "a z tile must not straddle a compaction boundary");
    constexpr int W = NC * FIELD_RB;  // lines per workgroup: line w = column w / 8, row w % 8
    using PL  = zdfft::Plan<L, E>;
    using LDS = zdfft::ColsInner<L, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int N = S.N;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    const int grp = blockIdx.y;
    const int r = w & (FIELD_RB - 1), x = blockIdx.x * NC + w / FIELD_RB;
    const int ky = ky0 + (grp * FIELD_RB + r) * S.ky_stride;
    if (S.prune & 2) {  // tile of identically-zero columns: the generator produced nothing, the y stage will not read it
        if (__syncthreads_and(column_is_zero(S, x > S.half ? x - N : x, ky))) return;
    }
    const char *src = reinterpret_cast<const char *>(Y + ((((long long) blockIdx.z * (nky / FIELD_RB) + grp) * L) * N) * FIELD_RB);
    const unsigned xb = (unsigned) (x * FIELD_RB + r) * 16u, rb = (unsigned) N * FIELD_RB * 16u;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = *reinterpret_cast<const cplx *>(src + ((unsigned) (t + T * e) * rb + xb));
        re[e] = v.x;
        im[e] = v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, twL);
    const FieldRow row = F.rows[kyloc0 / FIELD_RB + grp];  // workgroup-uniform: the row block
    const unsigned pos = ((unsigned) (x < row.split ? x : x - row.gap)) * FIELD_RB + r;
    const int Zq = 1 << F.lZq;
    int t2 = t;
    asm volatile("" : "+v"(t2));  // keep the store-address arithmetic after the FFT (register pressure)
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int z2  = t2 + T * e;
        const int dst = z2 >> F.lZq, zl = z2 & (Zq - 1);
        out[(long long) dst * F.chunk_elems + (long long) (zl * F.nfield + (int) blockIdx.z) * F.field_elems + (unsigned) row.base + pos]
            = cplx{re[e], im[e]};
    }
}

// k_yfft_f: y FFT of one of the three arrays the x stage consumes, built on the fly from the potentials (replaces
// LoadBlock's y shift + the Nyquist-row zeroing + half of Inverse2dFFT, block_array.cpp:466-504, zeldovich.cpp:644-658).
// The real fields are paired so that every potential is read by exactly ONE array (pairing qy with qz as the reference
// does would make E feed two arrays and be fetched twice):
//   a = 0, 1:  A_r = qx_r + i qy_r = (i kx - ky) E_r          (residue r = a of the pass)
//   a = 2:     C   = qz_0 + i qz_1 = i Z_0 - Z_1
//   rows y < N/2 (ky = y): as written.  Rows y > N/2 (ky = y - N): read row N - y at column N - x and use
//   E(-ky, -kx) = conj E(ky, kx), Z(-ky, -kx) = -conj Z(ky, kx):   A_r = (i kx + (N-y)) conj E_r,  C = -i conj Z_0 + conj Z_1
//   row y = N/2 and columns the zero rule kills: 0, not read.
// Output: ring[plane][a][row slot][x] in the single-rank block-store layout k_xfft reads.
//   grid: (3*N/W, 1, planes)   block: W*N/E
// One (plane pz of the launch, workgroup index id) unit of the y stage; all threads of the workgroup call it.
// TWO: the array is made of two potentials (a = 2, and every PLT array).  Compiled as two separate bodies so that the
// one-potential arrays (two thirds of the ZA workgroups) keep their E loads in flight without the register pressure of the
// second potential's batches.
template <int N, int E, int W, bool TWO>
__device__ __forceinline__ void yfft_f_unit_t(const FieldLayout &F, const StoreLayout &S, const cplx *__restrict__ tw,
                                              const cplx *__restrict__ store, int plane0, int ring_pitch, cplx *__restrict__ ring,
                                              int tile, int a, int pz, int unit, int t, int w, double *lds);

template <int N, int E, int W>
__device__ __forceinline__ void yfft_f_unit(const FieldLayout &F, const StoreLayout &S, const cplx *__restrict__ tw,
                                            const cplx *__restrict__ store, int plane0, int ring_pitch, cplx *__restrict__ ring,
                                            int id, int pz, int t, int w, double *lds) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, W>;
    constexpr int T = PL::T;
    constexpr int NT = N / W;
    int tile, a;
    ytile_of<NT, W>(id, tile, a);
    const int unit = pz * 3 * NT + id;
    if (F.nfield == 6 || a == 2)  // workgroup-uniform
        yfft_f_unit_t<N, E, W, true>(F, S, tw, store, plane0, ring_pitch, ring, tile, a, pz, unit, t, w, lds);
    else
        yfft_f_unit_t<N, E, W, false>(F, S, tw, store, plane0, ring_pitch, ring, tile, a, pz, unit, t, w, lds);
}

template <int N, int E, int W, bool TWO>
__device__ __forceinline__ void yfft_f_unit_t(const FieldLayout &F, const StoreLayout &S, const cplx *__restrict__ tw,
                                              const cplx *__restrict__ store, int plane0, int ring_pitch, cplx *__restrict__ ring,
                                              int tile, int a, int pz, [[maybe_unused]] int unit, int t, int w, double *lds) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, W>;
    constexpr int T = PL::T;
    const int x = tile * W + w, xm = (N - x) & (N - 1);
    const int zl = plane0 + pz;
#ifdef ZD_STAMPS
    if (zd_stamps && threadIdx.x == 0) {
        zd_stamps[(size_t) unit * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        zd_stamps[(size_t) unit * 8 + 7] = (unsigned long long) __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))  /* HW_ID */
                                           | ((unsigned long long) __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  /* XCC_ID */
    }
#endif
    ZD_STAMP(0, unit, false);
    const int kx = x > N / 2 ? x - N : x;
    // a tile none of whose columns has a live row (k_cutoff = 2: half of the tiles) is neither transformed nor written: the x
    // stage takes zeros for those columns (EpiConst::xdead_lo / hi, the same rule)
    if ((S.prune & PRUNE_YTILE) && __syncthreads_and(column_is_zero(S, kx, 0))) return;
    // potentials this array is made of.  ZA: E_a alone (a < 2), or (Z_0, Z_1).  PLT: the pairs (X, fX), (Y, Z), (fY, fZ) of
    // the fields X, Y, Z, fX, fY, fZ — every array is i P - Q like the ZA one of two
    const bool plt = F.nfield == 6;
    constexpr bool two = TWO;
    const int f0 = plt ? (a == 0 ? 0 : (a == 1 ? 1 : 4)) : (a == 2 ? 1 : 2 * a);
    const int f1 = plt ? (a == 0 ? 3 : (a == 1 ? 2 : 5)) : 3;
    const cplx *p0 = store + (long long) (zl * F.nfield + f0) * F.field_elems;
    const long long d01 = two ? (long long) (f1 - f0) * F.field_elems : 0;
    const int gmask = (1 << F.lG) - 1;
    double re[E], im[E];
    // Where the time went (in-kernel stamps, PPD = 4096: 59 k cycles per workgroup): 6 k waiting for the row records, 32 k for
    // the potentials in four dependent batches of 4 rows, 21 k in the transform (2 k of it the two table twiddles), 3 k for the
    // stores.  One workgroup fills a CU, so nothing else runs while it waits: every wait is taken ONCE, with everything the
    // workgroup will need already requested —
    //   (1) the row-block table (8 B x Hq/8 blocks) goes to LDS behind the transform's area: one coalesced load per thread
    //       instead of 16 scattered dependent ones; the twiddles of the later passes are requested at the same time;
    //   (2) all E rows of the first potential are loaded straight into re / im (64 registers: the ones the transform uses
    //       anyway), skipped rows from a clamped, always valid address; a second potential follows in two half batches;
    //   (3) combine in place, zero for skipped rows.
    FieldRow *rtab = reinterpret_cast<FieldRow *>(lds + LDS::SIZE);
    const int nblk = (S.Hq + FIELD_RB - 1) / FIELD_RB;
    for (int i = threadIdx.x; i < nblk; i += W * T) rtab[i] = F.rows[i];
    zdfft::TwSet<PL> twp;
    zdfft::load_twiddles<PL>(twp, t, tw);
    // which rows are skipped — integer arithmetic, done while the table is on its way.  column_is_zero's comparison
    // double(kx^2 + ky^2) * fund2 >= k2_cutoff is monotonic in the integer: c2 = the smallest integer that satisfies it
    unsigned skipm = 1u << ((N / 2 - t) / T);  // the Nyquist row y = N/2 is thread 0's (bit E/2; other threads: a bit they do not own)
    if (t != 0) skipm = 0;
    if (S.prune & 4) {
        int c2 = 0x7fffffff;
        if (S.k2_cutoff > 0) {
            int m = (int) (S.k2_cutoff / S.fund2);
            while (m > 0 && (double) m * S.fund2 >= S.k2_cutoff) m--;
            while ((double) m * S.fund2 < S.k2_cutoff) m++;
            c2 = m;
        }
        const int akx = kx < 0 ? -kx : kx, kx2 = kx * kx;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int y = t + T * e, kyp = y > N / 2 ? N - y : y;
            if (akx == S.kmax || kyp == S.kmax || kx2 + kyp * kyp >= c2) skipm |= 1u << e;
        }
    }
    __syncthreads();
    ZD_STAMP(1, unit, false);  // row records in
    // element of row slot e inside the (plane, field) image of its chunk (32 bits: an image is < 2^32 elements) + the chunk
    // (= source rank, several ranks only); skipped rows read element 0 of their chunk (their own record may describe an empty
    // row at the very end).  `tt` is the thread's line index through a register the compiler cannot see through: each phase
    // below gets its own, so that the address arithmetic of a later phase is not hoisted above the loads of an earlier one
    // (the scheduler did, and spilled 60-90 registers around the loads).
    auto locate = [&](int tt, int e) -> long long {
        const int y = tt + T * e;
        const bool mir = y > N / 2;
        int kyp = mir ? N - y : y;            // the half-space row that holds the data
        kyp = kyp < N / 2 ? kyp : N / 2 - 1;  // the Nyquist row is never used: any valid record
        const FieldRow row = rtab[(kyp >> F.lG) / FIELD_RB];
        const int xs = (mir && !ZD_TUNE(S.prune & 1024)) ? xm : x;  // bit 10: tuning ablation (aligned mirror reads)
        const unsigned off = (unsigned) row.base + (unsigned) (xs < row.split ? xs : xs - row.gap) * FIELD_RB + (unsigned) ((kyp >> F.lG) & (FIELD_RB - 1));
        return (long long) (kyp & gmask) * F.chunk_elems + (((skipm >> e) & 1u) ? 0u : off);
    };
    // (Measured and dropped: the tiles of the upper half issuing their rows in the opposite order, so that a tile and its mirror
    // tile — whose mirrored rows are the other's direct rows — ask for the same lines at the same point of their load phase:
    // y stage 724 -> 817 ms at PPD = 4096.)
    {
        int ta = t;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int e = 0; e < E; e++) {
            const cplx u = ld_stream(p0 + locate(ta, e), ZD_NTBIT(S, 64));
            re[e] = u.x;
            im[e] = u.y;
        }
    }
    if constexpr (TWO) {  // (u, v) = (Z_0, Z_1) or a PLT pair (P, Q):  i u - v,  mirrored -i conj u + conj v
#ifndef ZD_YHB
#define ZD_YHB (E / 2)
#endif
        constexpr int HB = ZD_YHB;
        const cplx *p1 = p0 + d01;
#pragma unroll
        for (int b = 0; b < E; b += HB) {
            cplx v[HB];
            int tb = t;
            asm volatile("" : "+v"(tb));
#pragma unroll
            for (int j = 0; j < HB; j++) v[j] = p1[locate(tb, b + j)];
            asm volatile("" : "+v"(tb));
#pragma unroll
            for (int j = 0; j < HB; j++) {
                const int e = b + j, y = tb + T * e;
                const double s = y > N / 2 ? -1.0 : 1.0;
                const double ur = re[e], ui = im[e];
                const bool skip = (skipm >> e) & 1u;
                re[e] = skip ? 0.0 : -ui - s * v[j].x;
                im[e] = skip ? 0.0 : s * ur - v[j].y;
            }
        }
    } else {  // u = E_r:  (i kx - ky) u,  mirrored (i kx + (N - y)) conj u
        const double dkx = (double) kx;
        int tc = t;
        asm volatile("" : "+v"(tc));
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int y = tc + T * e;
            const bool mir = y > N / 2;
            const double s = mir ? -1.0 : 1.0, dky = (double) (mir ? N - y : y);
            const double ur = re[e], ui = im[e];
            const bool skip = (skipm >> e) & 1u;
            re[e] = skip ? 0.0 : -s * (dky * ur + dkx * ui);
            im[e] = skip ? 0.0 : dkx * ur - dky * ui;
        }
    }
    if (ZD_TUNE(S.prune & 4096) && a == 2) return;                           // bit 12: tuning ablation (no x array)
    ZD_STAMP(2, unit, true);  // potentials in
    if (!ZD_TUNE(S.prune & 128)) zdfft::fft_line_tw<PL, LDS>(re, im, t, w, lds, tw, twp);  // bit 7: tuning ablation
    ZD_STAMP(3, unit, false);  // transformed
    if (ZD_TUNE(S.prune & 256) && re[0] != 123.456) return;                    // bit 8: tuning ablation (no stores)
    char *base = reinterpret_cast<char *>(ring + ((long long) (pz * 3 + a) * N) * ring_pitch);
    const unsigned xb = (unsigned) x * 16u, pb = (unsigned) ring_pitch * 16u;
    int t2 = t;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int y = t2 + T * e;
        const unsigned slot = y < N / 2 ? y : (y == N / 2 ? N / 2 : N / 2 + (N - y));
        if constexpr (N > 8192)  // one array plane of the ring exceeds 4 GB
            *reinterpret_cast<cplx *>(base + ((size_t) slot * pb + xb)) = cplx{re[e], im[e]};
        else
            st_stream(reinterpret_cast<cplx *>(base + (slot * pb + xb)), cplx{re[e], im[e]}, ZD_NTBIT(S, 32));
    }
    ZD_STAMP(4, unit, false);  // stores issued
    ZD_STAMP(5, unit, true);   // stores acknowledged (wave 0's)
}
//   grid: (3*N/W, 1, planes)   block: W*N/E
template <int N, int E, int W, int MINW = 1>
__global__ __launch_bounds__(W *N / E, MINW) void k_yfft_f(FieldLayout F, StoreLayout S, const cplx *__restrict__ tw,
                                                          const cplx *__restrict__ store, int plane0, int ring_pitch,
                                                          cplx *__restrict__ ring) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    yfft_f_unit<N, E, W>(F, S, tw, store, plane0, ring_pitch, ring, (int) blockIdx.x, (int) blockIdx.z, (int) threadIdx.x / W,
                         (int) threadIdx.x % W, lds);
}
// Persistent form: gridDim.x workgroups (a multiple of 8, so that a workgroup keeps its XCD class = linear index mod 8) walk
// the (plane, workgroup index) list with stride gridDim.x: the ring stores of one unit and the potential loads of the next
// are in flight together (as separate workgroups filling the CU, the next one's loads wait for the previous one's stores).
template <int N, int E, int W, int MINW = 1>
__global__ __launch_bounds__(W *N / E, MINW) void k_yfft_fp(FieldLayout F, StoreLayout S, const cplx *__restrict__ tw,
                                                           const cplx *__restrict__ store, int plane0, int ring_pitch,
                                                           cplx *__restrict__ ring, int nplanes) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int PER = 3 * (N / W);
    const int nwork = PER * nplanes;
#pragma unroll 1
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // per-iteration value: nothing that depends on the thread index is hoisted (and spilled)
        yfft_f_unit<N, E, W>(F, S, tw, store, plane0, ring_pitch, ring, work % PER, work / PER, tid / W, tid % W, lds);
    }
}

// ------------------------------------------------------------------------------------------------
// k_xfft: x FFT of ROWS rows x NA arrays per workgroup, then the WriteParticlesSlab epilogue.
//   grid: (N/ROWS, nplanes)   block: ROWS*NA*N/E
// passes over x of the epilogue: keep the field staging area of a workgroup <= 64 KB
constexpr int XFFT_NH(int N, int NA, int ROWS) {
    int nh = 1;
    while ((long long) ROWS * 2 * NA * N * 8 / nh > 65536 && nh < 16) nh *= 2;
    return nh;
}

// passes over x of k_xfft_seq's epilogue: the staged fields (2*NA*N/NH doubles) reuse the FFT's LDS (lds_dbl doubles)
constexpr int XFFT_SEQ_NH(int N, int NA, int lds_dbl) {
    int nh = 1;
    while (2 * NA * (N / nh) > lds_dbl && nh < 16) nh *= 2;
    return nh;
}

// the part of WriteParticlesSlab (src/output.cpp:86-203) that consumes the staged fields of x pass h:
// fld[(row*2*NA + 2a + {0,1})*NXH + x - h*NXH] -> records / density / running reductions
template <int N, int NA, int ROWS, int NXH, int NT>
__device__ __forceinline__ void xfft_consume(const EpiConst &ec, const double *__restrict__ fld, int h, int z, long long plane_rec0,
                                             char *__restrict__ records, float *__restrict__ density, double &ssq,
                                             MaxAbs32 &mx) {
    for (int i = threadIdx.x; i < ROWS * NXH; i += NT) {
            const int r = i / NXH, xl = i - r * NXH, xx = xl + h * NXH;
            const int yy = blockIdx.x * ROWS + r;
            const double *f = fld + (r * 2 * NA) * NXH + xl;
            auto finish = [&](long long pidx, int zz, const double (&pos)[3], const double (&vel)[3]) {
                // (rows and x ranges are not visited in index order here: ties by index.  The index tracked is local to the workgroup —
                // (plane of the pair, row, column) — and made global by k_xfft before the reduction: registers)
                max_track_any(mx, pos, ((zz != z ? ROWS : 0) + r) * N + xx);
                if (records) emit_record(records, pidx, ec, zz, yy, xx, pos, vel);
            };
            if constexpr (NA == 3) {
                // packed stores (zd_device.h PACK_*): no density field
                if (ec.pack == PACK_PLT3) {  // qx + i vx | qy + i qz | vy + i vz
                    const double pos[3] = {f[0], f[2 * NXH], f[3 * NXH]};
                    const double vel[3] = {f[1 * NXH] * ec.vnorm, f[4 * NXH] * ec.vnorm, f[5 * NXH] * ec.vnorm};
                    finish(plane_rec0 + (long long) yy * N + xx, z, pos, vel);
                } else {  // two planes, z and z + z_pair, from the three arrays
                    // PACK_ZAPAIR: (qy + i qz)_r0 | (qy + i qz)_r1 | qx_r0 + i qx_r1      PACK_ZAFIELD (ring of k_yfft_f):
                    // (qx + i qy)_r0 | (qx + i qy)_r1 | qz_r0 + i qz_r1
                    const bool fld3 = ec.pack == PACK_ZAFIELD;
#pragma unroll
                    for (int w2 = 0; w2 < 2; w2++) {
                        const double pos[3] = {f[(fld3 ? 2 * w2 : 4 + w2) * NXH], f[(fld3 ? 2 * w2 + 1 : 2 * w2) * NXH],
                                               f[(fld3 ? 4 + w2 : 2 * w2 + 1) * NXH]};
                        const double vel[3] = {pos[0] * ec.vnorm, pos[1] * ec.vnorm, pos[2] * ec.vnorm};
                        finish(2 * plane_rec0 + (long long) w2 * N * N + (long long) yy * N + xx, z + w2 * ec.z_pair, pos, vel);
                    }
                }
            } else {
                const double dens = f[0];
                const long long pidx = plane_rec0 + (long long) yy * N + xx;
                ssq += dens * dens;
                if (density) density[pidx] = (float) dens;
                if (NA >= 2) {
                    double pos[3], vel[3];
                    pos[0] = f[1 * NXH];
                    pos[1] = f[2 * NXH];
                    pos[2] = f[3 * NXH];
                    if (NA == 4) {
                        vel[0] = f[5 * NXH] * ec.vnorm;
                        vel[1] = f[6 * NXH] * ec.vnorm;
                        vel[2] = f[7 * NXH] * ec.vnorm;
                    } else {
                        vel[0] = pos[0] * ec.vnorm;
                        vel[1] = pos[1] * ec.vnorm;
                        vel[2] = pos[2] * ec.vnorm;
                    }
                    finish(pidx, z, pos, vel);
                }
            }
        }
}

// PLT3D: the instantiation for the PLT3 packing (records from line 1's registers); a separate kernel so that the ZA field-store
// instantiation keeps its 138 registers (with both branches in one body: 144, and 1 % on the default bench's x stage)
template <int N, int E, int NA, int ROWS, bool PLT3D = false>
__global__ __launch_bounds__(ROWS *NA *N / E, (ROWS * NA * N / E == 512 ? 4 : 1)) void k_xfft(StoreLayout S, EpiConst ec, const cplx *__restrict__ tw,
                                                         const cplx *__restrict__ data, int plane0,
                                                         int z_first, int z_step, char *__restrict__ records,
                                                         float *__restrict__ density, Reduce *__restrict__ red) {
    using PL = zdfft::Plan<N, E>;
    constexpr int WL = ROWS * NA;  // lines per workgroup
    using LDS = zdfft::LineInner<N, WL>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T, NT = WL * T;
    const int t = threadIdx.x % T, line = threadIdx.x / T;
    const int row = line / NA, a = line % NA;
    const int y  = blockIdx.x * ROWS + row;
    const int pl = plane0 + blockIdx.y;  // local plane index inside the store
    const cplx *src = data + row_offset(S, pl, a, y);
    double re[E], im[E];
    // the twiddles of the later passes are requested together with the row (one wait instead of three; zd_fft.h TwSet)
    zdfft::TwSet<PL> twp;
    zdfft::load_twiddles<PL>(twp, t, tw);
#pragma unroll
    for (int e = 0; e < E; e++) {
        // (dead columns of the field stores' ring are not even written: read element 0 of the row instead — one line for the
        // whole wave — and take zero; branch-free, like the y stage's skipped rows)
        const bool dead = x_is_dead(ec, t + T * e);
        const cplx v = ld_stream(src + (dead ? 0 : t + T * e), ZD_NTBIT(S, 4));
        re[e] = dead ? 0.0 : v.x;
        im[e] = dead ? 0.0 : v.y;
    }
    if (!ZD_TUNE(S.prune & 32)) zdfft::fft_line_tw<PL, LDS>(re, im, t, line, lds, tw, twp);  // bit 5: tuning ablation
    if (ZD_TUNE(S.prune & 64) && re[0] != 123.456) return;                           // bit 6: tuning ablation (no epilogue)

    // ---- WriteParticlesSlab (src/output.cpp:86-203) ----
    const int z = z_first + z_step * (int) blockIdx.y;
    double ssq = 0.0;
    MaxAbs32 mx;  // (column / workgroup-local tags; made lattice indices in front of the reduction)
    MaxAbs mg;
    const long long plane_rec0 = (long long) blockIdx.y * N * N;
    if constexpr (NA == 3 && !PLT3D) {
        if (ec.pack == PACK_ZAFIELD) {
            // Field-store ring: (qx + i qy)_r0 | (qx + i qy)_r1 | qz_r0 + i qz_r1.  The threads of lines 0 and 1 already
            // hold qx, qy of THEIR plane at their columns x = t + T*e; only the shared third array goes through LDS
            // (16 B per column instead of staging all six fields in NH rounds), and records leave straight from the
            // registers, consecutive lanes = consecutive records.
            double2 *cz = reinterpret_cast<double2 *>(lds);  // [row][x] = {qz_r0, qz_r1}
            if (a == 2) {
#pragma unroll
                for (int e = 0; e < E; e++) cz[row * N + t + T * e] = double2{re[e], im[e]};
            }
            __syncthreads();
            if (a < 2) {
                int t2 = t;
                asm volatile("" : "+v"(t2));
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int xx = t2 + T * e;
                    const double2 c = cz[row * N + xx];
                    const double pos[3] = {re[e], im[e], a ? c.y : c.x};
                    const double vel[3] = {pos[0] * ec.vnorm, pos[1] * ec.vnorm, pos[2] * ec.vnorm};
                    max_track(mx, pos, xx);  // (column only; the row's base is added below)
                    if (records)
                        emit_record(records, 2 * plane_rec0 + (long long) a * N * N + (long long) y * N + xx, ec, z + a * ec.z_pair, y, xx,
                                    pos, vel, ZD_NTBIT(S, 128));
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; j++) {
                mg.v[j]   = mx.v[j];
                mg.lin[j] = ((unsigned long long) (z + a * ec.z_pair) * N + (unsigned long long) y) * N + (unsigned) mx.tag[j];
            }
            xfft_reduce<NT, NA>(lds, red, ssq, mg);
            return;
        }
    }
    if constexpr (NA == 3 && PLT3D) {
        {
            // qx + i vx | qy + i qz | vy + i vz, ONE plane: the threads of line 1 hold qy, qz at their columns and write the
            // records; lines 0 and 2 hand {qx, vx} and {vy, vz} over through LDS ([row][x] and [ROWS + row][x], 16 B each) —
            // one exchange and one barrier instead of NH staging rounds of all six fields.
            double2 *st = reinterpret_cast<double2 *>(lds);
            if (a != 1) {
#pragma unroll
                for (int e = 0; e < E; e++) st[((a ? ROWS : 0) + row) * N + t + T * e] = double2{re[e], im[e]};
            }
            __syncthreads();
            if (a == 1) {
                int t2 = t;
                asm volatile("" : "+v"(t2));
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int xx = t2 + T * e;
                    const double2 c0 = st[row * N + xx], c2 = st[(ROWS + row) * N + xx];
                    const double pos[3] = {c0.x, re[e], im[e]};
                    const double vel[3] = {c0.y * ec.vnorm, c2.x * ec.vnorm, c2.y * ec.vnorm};
                    if (records) emit_record(records, plane_rec0 + (long long) y * N + xx, ec, z, y, xx, pos, vel);
                }
                // (max_disp in a loop of its own: inside the record loop it cost 36 spilled registers at PPD = 8192)
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int xx = t2 + T * e;
                    const double pos[3] = {st[row * N + xx].x, re[e], im[e]};
                    max_track(mx, pos, xx);  // (column only; the row's base is added below)
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; j++) {
                mg.v[j]   = mx.v[j];
                mg.lin[j] = ((unsigned long long) z * N + (unsigned long long) y) * N + (unsigned) mx.tag[j];
            }
            xfft_reduce<NT, NA>(lds, red, ssq, mg);
            return;
        }
    }
    // The unpacked fields go through LDS in NH passes over x so that the staging area stays <= 64 KB
    // (two workgroups per CU): pass h covers x in [h*N/NH, (h+1)*N/NH) = elements e in [h*E/NH, ...).
    constexpr int NH = XFFT_NH(N, NA, ROWS), NXH = N / NH, EH = E / NH;
    double *fld = lds;  // fld[(row*2*NA + 2a + {0,1})*NXH + x - h*NXH]
    int t2 = t;
    asm volatile("" : "+v"(t2));  // keep the staging-address arithmetic after the FFT (register pressure)
#pragma unroll
    for (int h = 0; h < NH; h++) {
        __syncthreads();
#pragma unroll
        for (int e2 = 0; e2 < EH; e2++) {
            const int e = h * EH + e2, xl = t2 + T * e - h * NXH;
            fld[(row * 2 * NA + 2 * a) * NXH + xl]     = re[e];
            fld[(row * 2 * NA + 2 * a + 1) * NXH + xl] = im[e];
        }
        __syncthreads();
        // (Writing 32-byte records by lane pairs — 16 B per lane, 1 KB of consecutive bytes per wave — was measured
        // 15 % SLOWER than one lane per record: the extra LDS reads cost more than the store pattern gains.)
        xfft_consume<N, NA, ROWS, NXH, NT>(ec, fld, h, z, plane_rec0, records, density, ssq, mx);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {  // workgroup-local index of xfft_consume -> lattice index
        const unsigned lt = (unsigned) mx.tag[j], w2 = lt / (unsigned) (ROWS * N), rr = (lt / (unsigned) N) % (unsigned) ROWS, xx = lt % (unsigned) N;
        mg.v[j]   = mx.v[j];
        mg.lin[j] = ((unsigned long long) (z + (int) w2 * ec.z_pair) * N + (unsigned long long) (blockIdx.x * ROWS + rr)) * N + xx;
    }
    xfft_reduce<NT, NA>(lds, red, ssq, mg);
}

// k_xfft_seq: the x pass of the field store with ONE row per workgroup: N/E threads transform its three arrays one after the
// other.  Built for PPD = 8192, where the three lines of a row do not fit a workgroup (1536 threads, 209 KB of LDS); since round 3
// also the form PPD = 4096 and 2048 run, because it leaves room for two (four) independent workgroups per CU.  The ring holds
// (qx + i qy)_r0 | (qx + i qy)_r1 | qz_r0 + i qz_r1, so the two delivered planes need {array 0, Re array 2} and
// {array 1, Im array 2}: array 2 is transformed first and kept in registers, then each of the other two is transformed
// and its plane's records are written at once — never more than two arrays' results are live.
//   grid: (N, nplanes)   block: N/E
template <int N, int E>
__global__ __launch_bounds__(N / E, 2) void k_xfft_seq(StoreLayout S, EpiConst ec, const cplx *__restrict__ tw,
                                                   const cplx *__restrict__ data, int plane0, int z_first, int z_step,
                                                   char *__restrict__ records, Reduce *__restrict__ red) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::LineInner<N, 1>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T, NT = T;
    const int t = threadIdx.x;
    const int y = blockIdx.x, pl = plane0 + blockIdx.y;
    // array 2 = qz_r0 + i qz_r1 goes first: its real part (plane r0, written first) stays in registers, its imaginary part waits in
    // LDS behind the transform's area.  ONE copy of the transform in a rolled loop over the arrays 2, 0, 1 (three inlined copies
    // with both arrays' results live spilled 28-78 registers).
    double *czi = lds + LDS::SIZE;
    double cr[E];
#pragma unroll
    for (int e = 0; e < E; e++) cr[e] = 0.0;
    const int z = z_first + z_step * (int) blockIdx.y;
    MaxAbs mx;
    const long long plane_rec0 = 2 * (long long) blockIdx.y * N * N;
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        const int a = it == 0 ? 2 : it - 1;
        const cplx *src = data + row_offset(S, pl, a, y);
        double re[E], im[E];
        int ta = t;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int e = 0; e < E; e++) {
            const cplx v = src[x_is_dead(ec, ta + T * e) ? 0 : ta + T * e];  // (see k_xfft)
            re[e] = v.x;
            im[e] = v.y;
        }
        int t3 = t;
        asm volatile("" : "+v"(t3));  // the dead flags are recomputed behind the loads rather than kept beside them (registers)
#pragma unroll
        for (int e = 0; e < E; e++)
            if (x_is_dead(ec, t3 + T * e)) re[e] = im[e] = 0.0;
        __syncthreads();  // the previous transform's last LDS reads are done
        zdfft::fft_line<PL, LDS>(re, im, t, 0, lds, tw);
        int t2 = t;
        asm volatile("" : "+v"(t2));
        if (it == 0) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                cr[e] = re[e];
                czi[t2 + T * e] = im[e];  // (a thread reads back only what it wrote itself)
            }
            continue;
        }
        // records straight from the registers (element e of thread t is x = t + T e; adjacent lanes write adjacent records),
        // as in k_xfft's field-store branch: no staging through LDS, no barriers
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xx = t2 + T * e;
            const double pos[3] = {re[e], im[e], a ? czi[xx] : cr[e]};
            const double vel[3] = {pos[0] * ec.vnorm, pos[1] * ec.vnorm, pos[2] * ec.vnorm};
            max_track(mx, pos, ((unsigned long long) (z + a * ec.z_pair) * N + (unsigned long long) y) * N + (unsigned long long) xx);
            if (records) emit_record(records, plane_rec0 + (long long) a * N * N + (long long) y * N + xx, ec, z + a * ec.z_pair, y, xx, pos, vel);
        }
    }
    __syncthreads();
    xfft_reduce<NT, 3>(lds, red, 0.0, mx);
}

// k_xfft_seq_plt: the one-row form for the PLT3 packing qx + i vx | qy + i qz | vy + i vz (ONE plane per store plane): array 0 first
// (kept in registers), array 2 (both parts wait in LDS behind the transform's area), then array 1, whose threads write the records.
// One rolled copy of the transform, as in k_xfft_seq.
//   grid: (N, nplanes)   block: N/E
// SPLIT2: vy stays in registers too and only vz waits in LDS (8 B per column: PPD = 4096 then needs 67 KB instead of 99, two
// workgroups per CU)
template <int N, int E, bool SPLIT2 = false>
__global__ __launch_bounds__(N / E, 2) void k_xfft_seq_plt(StoreLayout S, EpiConst ec, const cplx *__restrict__ tw,
                                                          const cplx *__restrict__ data, int plane0, int z_first, int z_step,
                                                          char *__restrict__ records, Reduce *__restrict__ red) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::LineInner<N, 1>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T, NT = T;
    const int t = threadIdx.x;
    const int y = blockIdx.x, pl = plane0 + blockIdx.y;
    double2 *c2s = reinterpret_cast<double2 *>(lds + LDS::SIZE);  // [x] = {vy, vz}; a thread reads back only what it wrote itself
    double *c2i = lds + LDS::SIZE;                                // SPLIT2: [x] = vz
    double c0r[E], c0i[E], c2r[SPLIT2 ? E : 1];
#pragma unroll
    for (int e = 0; e < E; e++) c0r[e] = c0i[e] = 0.0;
#pragma unroll
    for (int e = 0; e < (SPLIT2 ? E : 1); e++) c2r[e] = 0.0;
    const int z = z_first + z_step * (int) blockIdx.y;
    MaxAbs mx;
    const long long plane_rec0 = (long long) blockIdx.y * N * N;
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        const int a = it == 0 ? 0 : (it == 1 ? 2 : 1);
        const cplx *src = data + row_offset(S, pl, a, y);
        double re[E], im[E];
        int ta = t;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int e = 0; e < E; e++) {
            const cplx v = src[x_is_dead(ec, ta + T * e) ? 0 : ta + T * e];  // (see k_xfft)
            re[e] = v.x;
            im[e] = v.y;
        }
        int t3 = t;
        asm volatile("" : "+v"(t3));
#pragma unroll
        for (int e = 0; e < E; e++)
            if (x_is_dead(ec, t3 + T * e)) re[e] = im[e] = 0.0;
        __syncthreads();  // the previous transform's last LDS reads are done
        zdfft::fft_line<PL, LDS>(re, im, t, 0, lds, tw);
        int t2 = t;
        asm volatile("" : "+v"(t2));
        if (it == 0) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                c0r[e] = re[e];
                c0i[e] = im[e];
            }
            continue;
        }
        if (it == 1) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                if constexpr (SPLIT2) {
                    c2r[e] = re[e];
                    c2i[t2 + T * e] = im[e];
                } else
                    c2s[t2 + T * e] = double2{re[e], im[e]};
            }
            continue;
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xx = t2 + T * e;
            double2 c2;
            if constexpr (SPLIT2)
                c2 = double2{c2r[e], c2i[xx]};
            else
                c2 = c2s[xx];
            const double pos[3] = {c0r[e], re[e], im[e]};
            const double vel[3] = {c0i[e] * ec.vnorm, c2.x * ec.vnorm, c2.y * ec.vnorm};
            max_track(mx, pos, ((unsigned long long) z * N + (unsigned long long) y) * N + (unsigned long long) xx);
            if (records) emit_record(records, plane_rec0 + (long long) y * N + xx, ec, z, y, xx, pos, vel);
        }
    }
    __syncthreads();
    xfft_reduce<NT, 3>(lds, red, 0.0, mx);
}

// k_xfft_q2_plt: k_xfft_seq_plt for a store with plane-interleaved rows (StoreLayout::lq = 2, the fused PLT Z stage): a row of
// the store holds 4 planes side by side (64 bytes per column), a workgroup transforms the lines of TWO of them (w = thread % 2:
// lanes 2i, 2i + 1 read 32 contiguous bytes, the other half of every 64 belongs to the workgroup of the neighbouring pair, which
// runs next to it on the same XCD), ColsInner<N, 2> exchanges, records from the registers as there.  256 threads at N = 2048, two
// workgroups per CU.  Planes outside [plane0, plane0 + nplanes) of a pair are transformed but deliver nothing.
// The two workgroups of a plane group (h = 0, 1: planes 4q + 2h, 4q + 2h + 1) read the same 128-byte lines: they are made
// neighbours in the dispatch stream of ONE XCD (workgroups go to the XCDs round-robin by their linear index: row y = xcd mod 8,
// positions alternate h = 0, 1), so that the second one finds the lines in that XCD's L2 (as two distant workgroups every line was
// fetched from HBM twice: x stage of PPD=2048 PLT 201 ms against 134 on the plain rows).
//   grid: (2 N, plane groups touched)   block: 2 N/E
template <int N, int E, int NP = 2>
__global__ __launch_bounds__(NP * N / E, NP == 2 ? 2 : 1) void k_xfft_q2_plt(StoreLayout S, EpiConst ec, const cplx *__restrict__ tw,
                                                            const cplx *__restrict__ data, int plane0, int nplanes, int z_first,
                                                            int z_step, char *__restrict__ records, Reduce *__restrict__ red, int hdist) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, NP>;  // NP = 2 planes of the group per workgroup (product) or all 4 (A/B: ZD_XQ_NP4)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T, NT = NP * T;
    const int w = threadIdx.x % NP, t = threadIdx.x / NP;
    static_assert(N % 8 == 0, "rows are dealt to the 8 XCDs");
    int y, h = 0;
    if constexpr (NP == 2) {
        const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
        const int grp = pos / (2 * hdist), m = pos % (2 * hdist);  // hdist rows with h = 0, then the same rows with h = 1
        h = m / hdist;
        y = ((grp * hdist + m % hdist) << 3) + xcd;
    } else {
        y = blockIdx.x;
    }
    const int pl = 4 * ((plane0 >> 2) + (int) blockIdx.y) + 2 * h + w;  // this thread's plane of the store
    if (pl - w + NP - 1 < plane0 || pl - w >= plane0 + nplanes) return;  // (uniform) none of the workgroup's planes is asked for
    const bool active = pl >= plane0 && pl < plane0 + nplanes;
    const int pi = pl - plane0;                                      // its position among the delivered planes
    double *c2i = lds + LDS::SIZE;                                   // [x][w] = vz; a thread reads back only what it wrote itself
    double c0r[E], c0i[E], c2r[E];
#pragma unroll
    for (int e = 0; e < E; e++) c0r[e] = c0i[e] = c2r[e] = 0.0;
    const int z = z_first + z_step * pi;
    MaxAbs mx;
    const long long plane_rec0 = (long long) pi * N * N;
    const int sub = pl & 3;
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        const int a = it == 0 ? 0 : (it == 1 ? 2 : 1);
        // after the y stage a row holds, per pair of columns, [plane 0..3][column 0..1] (k_yfft): element (x, plane) at
        // 8 (x / 2) + 2 (plane % 4) + x % 2
        const cplx *src = data + row_offset(S, pl >> 2, a, y) + 2 * sub;
        double re[E], im[E];
        int ta = t;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int xl = x_is_dead(ec, ta + T * e) ? 0 : ta + T * e;  // (see k_xfft)
            const cplx v = src[((xl >> 1) << 3) + (xl & 1)];
            re[e] = v.x;
            im[e] = v.y;
        }
        int t3 = t;
        asm volatile("" : "+v"(t3));
#pragma unroll
        for (int e = 0; e < E; e++)
            if (x_is_dead(ec, t3 + T * e)) re[e] = im[e] = 0.0;
        __syncthreads();  // the previous transform's last LDS reads are done
        zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);
        int t2 = t;
        asm volatile("" : "+v"(t2));
        if (it == 0) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                c0r[e] = re[e];
                c0i[e] = im[e];
            }
            continue;
        }
        if (it == 1) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                c2r[e] = re[e];
                c2i[NP * (t2 + T * e) + w] = im[e];
            }
            continue;
        }
        if (active) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int xx = t2 + T * e;
                const double pos[3] = {c0r[e], re[e], im[e]};
                const double vel[3] = {c0i[e] * ec.vnorm, c2r[e] * ec.vnorm, c2i[NP * xx + w] * ec.vnorm};
                max_track(mx, pos, ((unsigned long long) z * N + (unsigned long long) y) * N + (unsigned long long) xx);
                if (records) emit_record(records, plane_rec0 + (long long) y * N + xx, ec, z, y, xx, pos, vel);
            }
        }
    }
    __syncthreads();
    xfft_reduce<NT, 3>(lds, red, 0.0, mx);
}

// k_xfft_two: the same x pass as two launches of one line per workgroup (PPD = 16384: a line alone takes the 1024 threads
// of a workgroup at 128 VGPRs, and holding a second line's results beside the transform spilled; PPD = 8192 with PLT: a
// record needs all three arrays).  Launch 0 transforms the arrays the records only READ, each over its own ring row;
// launch 1 transforms the remaining ones and writes the records straight from the registers, the rest from the rows
// launch 0 left (L2-resident: written a moment ago by the same XCD's neighbours).
//   ZA field ring  (qx + i qy)_r0 | (qx + i qy)_r1 | qz_r0 + i qz_r1:   launch 0: array 2 (grid z = 1); launch 1: arrays 0, 1
//                  = the two planes of the pair (grid z = 2)
//   PLT ring       qx + i vx | qy + i qz | vy + i vz (PACK_PLT3):        launch 0: arrays 0 and 2 (grid z = 2); launch 1:
//                  array 1 (grid z = 1), one plane
//   grid: (N, planes, z)   block: N/E
template <int N, int E, bool PLT>
// (PPD = 8192: 512 threads held to 128 registers — it took 130 — so that TWO workgroups fit a CU, 70 KB of LDS each)
__global__ __launch_bounds__(N / E, (N / E <= 512 ? 4 : 1)) void k_xfft_two(StoreLayout S, EpiConst ec, const cplx *__restrict__ tw, cplx *data, int emit,
                                                   int plane0, int z_first, int z_step, char *__restrict__ records,
                                                   Reduce *__restrict__ red) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::LineInner<N, 1>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T, NT = T;
    const int t = threadIdx.x;
    const int y = blockIdx.x, pl = plane0 + blockIdx.y;
    constexpr bool plt = PLT;
    const int a = plt ? (emit ? 1 : 2 * (int) blockIdx.z) : (emit ? (int) blockIdx.z : 2);
    cplx *row2 = data + row_offset(S, pl, 2, y);
    cplx *row0 = data + row_offset(S, pl, 0, y);
    cplx *self = data + row_offset(S, pl, a, y);
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const bool dead = x_is_dead(ec, t + T * e);  // (see k_xfft)
        const cplx v = self[dead ? 0 : t + T * e];
        re[e] = dead ? 0.0 : v.x;
        im[e] = dead ? 0.0 : v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, 0, lds, tw);
    int t2 = t;
    asm volatile("" : "+v"(t2));  // keep the address arithmetic of the epilogue after the FFT (register pressure)
    if (!emit) {
#pragma unroll
        for (int e = 0; e < E; e++) self[t2 + T * e] = cplx{re[e], im[e]};
        // max_disp of the displacement component(s) this launch holds in registers (launch 1 reads them back from the ring row and
        // tracks only its own two: tracking inside its record loop cost 40 spilled registers): PLT array 0 = qx + i vx -> axis 0;
        // ZA array 2 = qz_r0 + i qz_r1 -> axis 2 of the two planes of the pair
        if (!plt || a == 0) {
            MaxAbs32 m0, m1;
            constexpr int J = plt ? 0 : 2;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int xx = t2 + T * e;
                const bool g0 = fabs(re[e]) > fabs(m0.v[J]);
                m0.v[J]   = g0 ? re[e] : m0.v[J];
                m0.tag[J] = g0 ? xx : m0.tag[J];
                if constexpr (!plt) {
                    const bool g1 = fabs(im[e]) > fabs(m1.v[J]);
                    m1.v[J]   = g1 ? im[e] : m1.v[J];
                    m1.tag[J] = g1 ? xx : m1.tag[J];
                }
            }
            const int z0 = z_first + z_step * (int) blockIdx.y;
            MaxAbs mg;
            mg.v[J]   = m0.v[J];
            mg.lin[J] = ((unsigned long long) z0 * N + (unsigned long long) y) * N + (unsigned) m0.tag[J];
            if constexpr (!plt) {  // the plane of the second residue: z0 + z_pair (later in z: it only wins with a larger magnitude)
                const unsigned long long l1 = ((unsigned long long) (z0 + ec.z_pair) * N + (unsigned long long) y) * N + (unsigned) m1.tag[J];
                if (max_better(m1.v[J], l1, mg.v[J], mg.lin[J])) {
                    mg.v[J]   = m1.v[J];
                    mg.lin[J] = l1;
                }
            }
            __syncthreads();
            xfft_reduce<NT, 3>(lds, red, 0.0, mg);
        }
        return;
    }
    const int z = z_first + z_step * (int) blockIdx.y + (plt ? 0 : a * ec.z_pair);
    MaxAbs32 mx;
    const long long rec0 = plt ? (long long) blockIdx.y * N * N + (long long) y * N
                               : 2 * (long long) blockIdx.y * N * N + (long long) a * N * N + (long long) y * N;
    // max_disp: the two components this thread holds in registers, in a loop of their own in front of the records (registers only; the
    // third component was tracked by launch 0, which held it in registers)
    constexpr int JA = plt ? 1 : 0, JB = plt ? 2 : 1;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int xx = t2 + T * e;
        const bool ga = fabs(re[e]) > fabs(mx.v[JA]), gb = fabs(im[e]) > fabs(mx.v[JB]);
        mx.v[JA]   = ga ? re[e] : mx.v[JA];
        mx.tag[JA] = ga ? xx : mx.tag[JA];
        mx.v[JB]   = gb ? im[e] : mx.v[JB];
        mx.tag[JB] = gb ? xx : mx.tag[JB];
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int xx = t2 + T * e;
        const cplx c2 = row2[xx];
        double pos[3], vel[3];
        if constexpr (plt) {
            const cplx c0 = row0[xx];
            pos[0] = c0.x; pos[1] = re[e]; pos[2] = im[e];
            vel[0] = c0.y * ec.vnorm; vel[1] = c2.x * ec.vnorm; vel[2] = c2.y * ec.vnorm;
        } else {
            pos[0] = re[e]; pos[1] = im[e]; pos[2] = a ? c2.y : c2.x;
            vel[0] = pos[0] * ec.vnorm; vel[1] = pos[1] * ec.vnorm; vel[2] = pos[2] * ec.vnorm;
        }
        if (records) emit_record(records, rec0 + xx, ec, z, y, xx, pos, vel);
    }
    __syncthreads();
    MaxAbs mg;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        mg.v[j]   = mx.v[j];
        mg.lin[j] = ((unsigned long long) z * N + (unsigned long long) y) * N + (unsigned) mx.tag[j];
    }
    xfft_reduce<NT, 3>(lds, red, 0.0, mg);
}

// ------------------------------------------------------------------------------------------------
// f_NL kernels (ZeldovichXY_Phi + the forward transforms, src/zeldovich.cpp:699-790, 116-135, 324-326).
// A forward (sign -1) transform is conj o inverse o conj, so the same register-resident engine serves.

// M(k) table by integer |k|^2: infer_Tk / primordial_power (src/power_spectrum.cpp:263-274)
template <bool PLAW>
__global__ void k_fnl_table(GenConst g, int n, double *__restrict__ tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double k2 = (double) i * g.fundamental2;
    const double kmag = sqrt(k2);
    double Tk = 1.0;
    if (kmag > 0.0) Tk = sqrt(pk_power<PLAW>(g, k2) / (g.primordial_norm * exp(log(kmag) * g.n_s)));
    if (k2 == 0.0) k2 = 1.0;
    tab[i] = g.fnl_pre * Tk * k2 / g.fnl_den;
}

// rows of the (narray = 1) store: inverse x FFT -> phi(x) -> (phi + f_NL phi^2)/N^3 -> forward x FFT, in place
template <int N, int E, int ROWS>
__global__ __launch_bounds__(ROWS *N / E) void k_xphi(StoreLayout S, double f_NL, double inv_ppd3, const cplx *__restrict__ tw,
                                                     cplx *__restrict__ data) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::LineInner<N, ROWS>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int t = threadIdx.x % T, row = threadIdx.x / T;
    const int y = blockIdx.x * ROWS + row, pl = blockIdx.y;
    cplx *p = data + row_offset(S, pl, 0, y);
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = p[t + T * e];
        re[e] = v.x;
        im[e] = v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, row, lds, tw);
#pragma unroll
    for (int e = 0; e < E; e++) {
        const double phi = re[e];  // real(AZYX(slab,0,...)), zeldovich.cpp:753
        re[e] = (phi + f_NL * phi * phi) * inv_ppd3;
        im[e] = 0.0;
    }
    __syncthreads();
    zdfft::fft_line<PL, LDS>(re, im, t, row, lds, tw);  // input is real: conj is the identity
    int t2 = t;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int e = 0; e < E; e++) p[t2 + T * e] = cplx{re[e], -im[e]};
}

// forward y FFT in place on the store (all rows are data: no Nyquist zeroing, no pruning)
template <int N, int E, int W>
__global__ __launch_bounds__(W *N / E) void k_yfwd(StoreLayout S, const cplx *__restrict__ tw, cplx *__restrict__ data) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    cplx *base = data + blockIdx.x * W + w;
    const int zl = blockIdx.z;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = base[row_offset(S, zl, 0, t + T * e)];
        re[e] = v.x;
        im[e] = -v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);
    int t2 = t;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int e = 0; e < E; e++) base[row_offset(S, zl, 0, t2 + T * e)] = cplx{re[e], -im[e]};
}

// forward z FFT of this rank's half-space rows: gathers a column tile across all N planes — plane z sits in chunk z / Zq
// (the rank that did its XY stage), local plane z % Zq, at this rank's row slot — and writes PhiK[row slot][kz][x] (what
// LoadBlockForward + ForwardFFT_Yonly hand to LoadPlane; with G ranks the slot of ky is ky / G)
template <int N, int E, int W>
__global__ __launch_bounds__(W *N / E) void k_zfwd(StoreLayout S, int lZq, const cplx *__restrict__ tw, const cplx *__restrict__ data,
                                                  cplx *__restrict__ phik) {
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    const int x = blockIdx.x * W + w, ky = blockIdx.y;  // ky: row slot
    const int zmask = (1 << lZq) - 1;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int z = t + T * e;
        const cplx v = data[store_offset(S, z >> lZq, z & zmask, 0, ky) + x];
        re[e] = v.x;
        im[e] = -v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);
    int t2 = t;
    asm volatile("" : "+v"(t2));
#pragma unroll
    for (int e = 0; e < E; e++) phik[((long long) ky * N + (t2 + T * e)) * N + x] = cplx{re[e], -im[e]};
}

#ifdef ZD_TESTING
// ------------------------------------------------------------------------------------------------
// test kernels: batches of independent lines through the two LDS layouts
template <int N, int E, int W>
__global__ __launch_bounds__(W *N / E) void k_test_fft_cols(const cplx *__restrict__ tw, const cplx *__restrict__ in,
                                                           cplx *__restrict__ out, long long lines) {
    // data layout [n][lines] (line index contiguous): the strided-axis situation of the y/z passes
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::ColsInner<N, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    const long long col = (long long) blockIdx.x * W + w;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = in[(long long) (t + T * e) * lines + col];
        re[e] = v.x;
        im[e] = v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);
#pragma unroll
    for (int e = 0; e < E; e++) out[(long long) (t + T * e) * lines + col] = cplx{re[e], im[e]};
}
template <int N, int E, int W>
__global__ __launch_bounds__(W *N / E) void k_test_fft_lines(const cplx *__restrict__ tw, const cplx *__restrict__ in,
                                                            cplx *__restrict__ out, long long lines) {
    // data layout [lines][n]: contiguous lines (x pass)
    using PL  = zdfft::Plan<N, E>;
    using LDS = zdfft::LineInner<N, W>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = PL::T;
    const int t = threadIdx.x % T, w = threadIdx.x / T;
    const long long line = (long long) blockIdx.x * W + w;
    double re[E], im[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const cplx v = in[line * N + t + T * e];
        re[e] = v.x;
        im[e] = v.y;
    }
    zdfft::fft_line<PL, LDS>(re, im, t, w, lds, tw);
#pragma unroll
    for (int e = 0; e < E; e++) out[line * N + t + T * e] = cplx{re[e], im[e]};
}

#endif  // ZD_TESTING
#ifdef ZD_TUNING
__global__ void k_copy16(const uint4 *__restrict__ in, uint4 *__restrict__ out, long long n) {
    long long i      = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = (long long) gridDim.x * blockDim.x;
    for (; i < n; i += s) out[i] = in[i];
}

#endif

// ================================================================================================
// launchers (C++ linkage inside the library; the C ABI lives in zd_capi.cpp)


namespace zd {

int zfft_tile_width(int L);
int zfft_fields_tile_columns(int L);

// z rows walked by one generator thread: 16; 4 for the short composite z lines (L = 108 = 4 * 27: PPD = 6912 on ONE GPU);
// 4, 2 or 1 for the arbitrary lengths of the any-PPD path (zd_kernels_any.hip)
#ifdef ZD_GEN_ZR_FORCE  // experiment (make variant): a shorter walk = smaller generator tiles
static int gen_zr(int L) { return L % ZD_GEN_ZR_FORCE == 0 ? ZD_GEN_ZR_FORCE : (L % 2 == 0 ? 2 : 1); }
#else
static int gen_zr(int L) { return L % GEN_ZR == 0 ? GEN_ZR : (L % 4 == 0 ? 4 : (L % 2 == 0 ? 2 : 1)); }
#endif

template <int ZR, int NJ, bool PLT, bool PLAW>
static int launch_gen_z(const GenConst &g, const GenJumps &J, const JobList &jobs, const StoreLayout &S, int ky0, int nky,
                        int nrows, int L, int residue, int residue2, const void *twN, void *Y, hipStream_t st) {
    const int N = g.N;
    dim3 grid((N + GEN_BX - 1) / GEN_BX, L / ZR, nrows), block(GEN_BX);
    const int zw = zfft_tile_width(L) > 0 ? zfft_tile_width(L) : 16;  // (composite L: any width is a safe neighbourhood)
    hipLaunchKernelGGL((k_gen<ZR, NJ, PLT, PLAW>), grid, block, 0, st, g, J, jobs, S, zw, ky0, nky,
                       L, residue, residue2, (const cplx *) twN, (cplx *) Y);
    ZD_LAUNCH_CHECK();
    return 0;
}
template <int NJ, bool PLT, bool PLAW>
static int launch_gen_t(const GenConst &g, const GenJumps &J, const JobList &jobs, const StoreLayout &S, int ky0, int nky,
                        int nrows, int L, int residue, int residue2, const void *twN, void *Y, hipStream_t st) {
    if (gen_zr(L) == GEN_ZR) return launch_gen_z<GEN_ZR, NJ, PLT, PLAW>(g, J, jobs, S, ky0, nky, nrows, L, residue, residue2, twN, Y, st);
    // the short walks exist for the field stores (their ky = 0 row: 4 ZA jobs, 6 PLT jobs) and for the reference arrays of the
    // any-PPD path (1, 4 or 7 jobs)
    if constexpr ((NJ == 4 && !PLT) || NJ == 6) {
        if (gen_zr(L) == 4 && pack_is_fields(jobs.pack))
            return launch_gen_z<4, NJ, PLT, PLAW>(g, J, jobs, S, ky0, nky, nrows, L, residue, residue2, twN, Y, st);
    }
    if constexpr (NJ == 1 || (NJ == 4 && !PLT) || (NJ == 7 && PLT)) {
        if (jobs.pack == PACK_NONE) {
            if (gen_zr(L) == 4) return launch_gen_z<4, NJ, PLT, PLAW>(g, J, jobs, S, ky0, nky, nrows, L, residue, residue2, twN, Y, st);
            if (gen_zr(L) == 2) return launch_gen_z<2, NJ, PLT, PLAW>(g, J, jobs, S, ky0, nky, nrows, L, residue, residue2, twN, Y, st);
            if (gen_zr(L) == 1) return launch_gen_z<1, NJ, PLT, PLAW>(g, J, jobs, S, ky0, nky, nrows, L, residue, residue2, twN, Y, st);
        }
    }
    return 2;
}
// (tuning library: ZD_GEN_NO_KZPAIR=1 keeps the unpaired PLT generator for A/B timings)
static bool kz_pair_off() {
#ifdef ZD_TUNING
    static const bool off = getenv("ZD_GEN_NO_KZPAIR") != nullptr;
    return off;
#else
    return false;
#endif
}
template <int ZR, int KIND, bool PLAW>
static int launch_genf_z(const GenConst &g, const GenJumps &J, const StoreLayout &S, int ky0, int kyl0, int nky, int nrows,
                         int L, int residue, int residue2, const void *twN, void *Y, unsigned *tile_ctr, int max_wgs,
                         hipStream_t st) {
    const int N = g.N;
    constexpr bool za = KIND == GENF_DENS || KIND == GENF_ZA || KIND == GENF_ZAP || KIND == GENF_ZAF || KIND == GENF_ZAFD;
#ifdef ZD_TUNING
    static const bool mirror_off = getenv("ZD_GEN_NO_MIRROR") != nullptr;
#else
    constexpr bool mirror_off = false;
#endif
    const bool mirror = za && (!mirror_off || KIND == GENF_ZAF || KIND == GENF_ZAFD);  // the field store's blocked layout exists in the mirror form only
    // the mirror form folds the second residue of a pass as (-1)^k1 times the first: it must be residue + R/2
    if (mirror && (KIND == GENF_ZAP || KIND == GENF_ZAF || KIND == GENF_ZAFD) && residue2 != residue + (N / L) / 2) return 2;
    const bool blk = (mirror && (KIND == GENF_ZAF || KIND == GENF_ZAFD)) || KIND == GENF_PLTF;
    const int xw = blk ? GEN_BX / FIELD_RB : GEN_BX;
    const int gx = ((mirror ? N / 2 + 1 : N) + xw - 1) / xw;
    // PLT kinds: the kz mirror folded in (genf_tile_kz, k_genf<.., true>) where the generator paces the Z stage — more than two
    // passes.  At R <= 2 (PPD = 2048 PLT, BASELINE C3) the stage is bound by HBM bytes and the unpaired kernel, three workgroups per
    // CU, is the faster one (Z stage 241 against 250 ms)
    const int kzpair = (!mirror && (KIND == GENF_PLTN || KIND == GENF_PLTF) && kz_chunks(L, ZR) > 0 && N / L > 2 && !kz_pair_off()) ? 1 : 0;
    const long long ntiles = (long long) gx * (kzpair ? kz_chunks(L, ZR) : L / ZR) * (blk ? nky / FIELD_RB : nrows);
    dim3 grid((unsigned) std::min<long long>(ntiles, max_wgs)), block(GEN_BX);
    const size_t shmem = sizeof(double) * (size_t) (g.genf_n + 2);
    if constexpr (za) {
        if (mirror) {
            hipLaunchKernelGGL((k_genf<ZR, KIND, PLAW, true>), grid, block, shmem, st, g, J, S,
                               (KIND == GENF_ZAF || KIND == GENF_ZAFD) ? zfft_fields_tile_columns(L) : zfft_tile_width(L), ky0, kyl0,
                               nky, nrows, L, residue, residue2, (const cplx *) twN, (cplx *) Y, tile_ctr);
            ZD_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (KIND == GENF_PLTN || KIND == GENF_PLTF) {
        if (kzpair) {
            hipLaunchKernelGGL((k_genf<ZR, KIND, PLAW, true>), grid, block, shmem, st, g, J, S,
                               KIND == GENF_PLTF ? zfft_fields_tile_columns(L) : zfft_tile_width(L), ky0, kyl0, nky,
                               nrows, L, residue, residue2, (const cplx *) twN, (cplx *) Y, tile_ctr);
            ZD_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (KIND == GENF_ZAFD) {
        return 2;  // (mirror form only)
    } else {
        hipLaunchKernelGGL((k_genf<ZR, KIND, PLAW, false>), grid, block, shmem, st, g, J, S,
                           KIND == GENF_PLTF ? zfft_fields_tile_columns(L) : zfft_tile_width(L), ky0, kyl0, nky,
                           nrows, L, residue, residue2, (const cplx *) twN, (cplx *) Y, tile_ctr);
        ZD_LAUNCH_CHECK();
        return 0;
    }
}
template <int KIND, bool PLAW>
static int launch_genf_t(const GenConst &g, const GenJumps &J, const StoreLayout &S, int ky0, int kyl0, int nky, int nrows,
                         int L, int residue, int residue2, const void *twN, void *Y, unsigned *tile_ctr, int max_wgs,
                         hipStream_t st) {
    if (gen_zr(L) == GEN_ZR)
        return launch_genf_z<GEN_ZR, KIND, PLAW>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
    if constexpr (KIND == GENF_ZAF || KIND == GENF_PLTF || KIND == GENF_ZAFD) {  // the short walk of the field stores (z lines of 108)
        if (gen_zr(L) == 4)
            return launch_genf_z<4, KIND, PLAW>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
    }
    if constexpr (KIND == GENF_DENS || KIND == GENF_ZA || KIND == GENF_PLT) {  // reference arrays of the any-PPD path
        if (gen_zr(L) == 4)
            return launch_genf_z<4, KIND, PLAW>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
        if (gen_zr(L) == 2)
            return launch_genf_z<2, KIND, PLAW>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
    }
    return 2;
}
template <int KIND>
static int launch_genf_k(const GenConst &g, const GenJumps &J, const StoreLayout &S, int ky0, int kyl0, int nky, int nrows,
                         int L, int residue, int residue2, const void *twN, void *Y, unsigned *tile_ctr, int max_wgs,
                         hipStream_t st) {
    if (g.is_powerlaw)
        return launch_genf_t<KIND, true>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
    return launch_genf_t<KIND, false>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
}
// the job orders k_genf writes (= the orders zd_plan builds)
static int genf_kind(const JobList &jobs, bool plt) {
    static const int std7[7] = {JOB_A_SELF, JOB_A_TWIN, JOB_B_SELF, JOB_B_TWIN, JOB_C_BOTH, JOB_D_SELF, JOB_D_TWIN};
    static const int zap[6]  = {JOB_B_SELF, JOB_B_TWIN, JOB_B_SELF, JOB_B_TWIN, JOB_FX, JOB_FX};
    static const int pln[6]  = {JOB_XV_SELF, JOB_XV_TWIN, JOB_B_SELF, JOB_B_TWIN, JOB_D_SELF, JOB_D_TWIN};
    static const int zaf[4]  = {JOB_E, JOB_Z, JOB_E, JOB_Z};
    static const int plf[6]  = {JOB_PX, JOB_PY, JOB_PZ, JOB_PFX, JOB_PFY, JOB_PFZ};
    static const int zafd[6] = {JOB_E, JOB_Z, JOB_E, JOB_Z, JOB_DENS, JOB_DENS};
    auto same = [&](const int *ref, int n) {
        if (jobs.n != n) return false;
        for (int j = 0; j < n; j++)
            if (jobs.kind[j] != ref[j]) return false;
        return true;
    };
    if (jobs.n == 1 && jobs.kind[0] == JOB_DENS) return GENF_DENS;
    if (jobs.pack == PACK_ZAPAIR) return (!plt && same(zap, 6)) ? GENF_ZAP : -1;
    if (jobs.pack == PACK_ZAFIELD) return (!plt && same(zaf, 4)) ? GENF_ZAF : ((!plt && same(zafd, 6)) ? GENF_ZAFD : -1);
    if (jobs.pack == PACK_PLTFIELD) return (plt && same(plf, 6)) ? GENF_PLTF : -1;
    if (jobs.pack == PACK_PLT3) return (plt && same(pln, 6)) ? GENF_PLTN : -1;
    if (!plt && same(std7, 4)) return GENF_ZA;
    if (plt && same(std7, 7)) return GENF_PLT;
    return -1;
}
int launch_gen(const GenConst &g, const GenJumps &J, const JobList &jobs, const StoreLayout &S, int ky0, int nky, int L,
               int residue, int residue2, const void *twN, void *Y, unsigned *tile_ctr, int max_wgs, hipStream_t st) {
    if (gen_zr(L) == 0) return 2;
#ifdef ZD_TUNING
    static const bool force_general = getenv("ZD_GEN_GENERAL") != nullptr;
#else
    constexpr bool force_general = false;
#endif
    // rows ky >= 1 of a production run go through k_genf; everything else through the general kernel
    const int kind = genf_kind(jobs, g.qPLT != 0);
    const int zr = gen_zr(L);
    const bool ref_kind = kind == GENF_DENS || kind == GENF_ZA || kind == GENF_PLT;
    const bool zr_ok = zr == GEN_ZR || (zr == 4 && (kind == GENF_ZAF || kind == GENF_ZAFD || kind == GENF_PLTF || ref_kind)) || (zr == 2 && ref_kind);
    const bool fast = g.genf_tab && tile_ctr && !g.phik && !g.gen_phi && !g.qonemode && !g.v1dev && !ZD_TUNE(g.ablate & 15) && !force_general
                      && kind >= 0 && zr_ok;
    int general_rows = nky;
    if (fast) {
        general_rows = ky0 == 0 ? 1 : 0;  // the ky = 0 plane (conjugate "loser" modes) stays with k_gen
        const int kyl0 = general_rows, nrows = nky - general_rows;
        if (nrows > 0) {
            int rc = 2;
#define FCASE(K) \
    if (kind == K) rc = launch_genf_k<K>(g, J, S, ky0, kyl0, nky, nrows, L, residue, residue2, twN, Y, tile_ctr, max_wgs, st);
            FCASE(GENF_DENS)
            FCASE(GENF_ZA)
            FCASE(GENF_PLT)
            FCASE(GENF_ZAP)
            FCASE(GENF_PLTN)
            FCASE(GENF_ZAF)
            FCASE(GENF_PLTF)
            FCASE(GENF_ZAFD)
#undef FCASE
            if (rc) return rc;
        }
        if (general_rows == 0) return 0;
    }
#define GCASE(nj, plt)                                                                                                      \
    if (jobs.n == nj && (g.qPLT != 0) == plt) {                                                                             \
        if (g.is_powerlaw)                                                                                                  \
            return launch_gen_t<nj, plt, true>(g, J, jobs, S, ky0, nky, general_rows, L, residue, residue2, twN, Y, st);      \
        return launch_gen_t<nj, plt, false>(g, J, jobs, S, ky0, nky, general_rows, L, residue, residue2, twN, Y, st);         \
    }
    GCASE(1, false)
    GCASE(1, true)
    GCASE(4, false)
    GCASE(7, true)
    GCASE(6, false)
    GCASE(6, true)
#undef GCASE
    return 2;
}

// (x, y)-interpolated eigenmode lines of the rows ky0, ky0 + ky_stride, ... of a slab (GenConst::eig_lines)
int launch_eig_lines(const GenConst &g, int ky0, int ky_stride, int nrows, void *lines, hipStream_t st) {
    dim3 grid((g.N + 255) / 256, nrows, (unsigned) g.eig_ppd / 2 + 1), block(256);
    hipLaunchKernelGGL(k_eig_lines, grid, block, 0, st, g, ky0, ky_stride, nrows, (double *) lines);
    ZD_LAUNCH_CHECK();
    return 0;
}

int launch_pk_table(const GenConst &g, int n, void *tab, hipStream_t st) {
    dim3 grid((n + 255) / 256), block(256);
    if (g.is_powerlaw)
        hipLaunchKernelGGL(k_pk_table<true>, grid, block, 0, st, g, n, (double2 *) tab);
    else
        hipLaunchKernelGGL(k_pk_table<false>, grid, block, 0, st, g, n, (double2 *) tab);
    ZD_LAUNCH_CHECK();
    return 0;
}

#ifdef ZD_TESTING
int launch_test_modes(const GenConst &g, long long n, const int *kxyz, uint64_t *draws, double *D, hipStream_t st) {
    dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_test_modes, grid, block, 0, st, g, n, kxyz, draws, D);
    ZD_LAUNCH_CHECK();
    return 0;
}

int launch_test_modes_table(const GenConst &g, long long n, const int *kxyz, double *out, hipStream_t st) {
    if (!g.genf_tab) return 2;
    dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    const size_t shmem = sizeof(double) * (size_t) (g.genf_n + 2);
    if (g.is_powerlaw)
        hipLaunchKernelGGL(k_test_modes_table<true>, grid, block, shmem, st, g, n, kxyz, out);
    else
        hipLaunchKernelGGL(k_test_modes_table<false>, grid, block, shmem, st, g, n, kxyz, out);
    ZD_LAUNCH_CHECK();
    return 0;
}

#endif  // ZD_TESTING
template <int L, int E, int W>
static int launch_zfft_t(const JobList &jobs, const StoreLayout &S, int ky0, int kyloc0, int nky, int Zq, const void *Y,
                         const void *twL, void *out, hipStream_t st) {
    constexpr int threads = W * L / E;
    const size_t shmem = sizeof(double) * zdfft::ColsInner<L, W>::SIZE;
    set_dyn_lds<k_zfft<L, E, W>>(shmem);
    dim3 grid(S.N / W, nky, jobs.n), block(threads);
    hipLaunchKernelGGL((k_zfft<L, E, W>), grid, block, shmem, st, jobs, S, ky0, kyloc0, nky, Zq, (const cplx *) Y,
                       (const cplx *) twL, (cplx *) out);
    ZD_LAUNCH_CHECK();
    return 0;
}

int launch_zfft(int L, const JobList &jobs, const StoreLayout &S, int ky0, int kyloc0, int nky, int Zq, const void *Y,
                const void *twL, void *out, hipStream_t st) {
#define ZCASE(l, e, w) \
    case l: return launch_zfft_t<l, e, w>(jobs, S, ky0, kyloc0, nky, Zq, Y, twL, out, st);
    switch (L) {
        ZCASE(32, 16, 32)
        ZCASE(64, 16, 32)
        ZCASE(128, 16, 32)
        ZCASE(256, 16, 16)
        ZCASE(512, 16, 16)
        ZCASE(1024, 16, 8)
        ZCASE(2048, 16, 8)
        ZCASE(4096, 16, 4)
    }
#undef ZCASE
    fprintf(stderr, "zeldovich_hip: unsupported z-FFT length %d (power of two in [32,4096] required)\n", L);
    return 2;
}
int zfft_tile_width(int L) {
    switch (L) {
        case 32: case 64: case 128: return 32;
        case 256: case 512: return 16;
        case 1024: case 2048: return 8;
        case 4096: return 4;
        case 8192: return 2;
        case 16384: return 1;
    }
    return 0;
}

template <int N, int E, int W>
static int launch_yfft_t(const StoreLayout &S, int nplanes, const void *tw, void *data, hipStream_t st) {
    constexpr int threads = W * N / E;
    const size_t shmem = sizeof(double) * zdfft::ColsInner<N, W>::SIZE;
    set_dyn_lds<k_yfft<N, E, W, 1, false>>(shmem);
    set_dyn_lds<k_yfft<N, E, W, 1, true>>(shmem);
    // (lq = 2 with W a multiple of 8 only: the tile is then whole 128-byte lines per row, which the kernel writes back
    // [plane][column] — the order k_xfft_q2_plt reads)
    if (S.lq && (S.lq != 2 || W % 8 != 0 || !S.one_block || nplanes % (1 << S.lq) || (unsigned long long) S.pitch * N * 16ull >= (1ull << 32))) {
        fprintf(stderr, "zeldovich_hip: plane-interleaved rows need the single-rank store and whole plane groups\n");
        return 2;
    }
    dim3 grid((N << S.lq) / W, S.narray, nplanes >> S.lq), block(threads);
    if (S.one_block) {
        hipLaunchKernelGGL((k_yfft<N, E, W, 1, true>), grid, block, shmem, st, S, (const cplx *) tw, (cplx *) data);
        ZD_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL((k_yfft<N, E, W, 1, false>), grid, block, shmem, st, S, (const cplx *) tw, (cplx *) data);
        ZD_LAUNCH_CHECK();
    }
    return 0;
}
template <int L, int E, int NC>
static int launch_zfft_f_t(const FieldLayout &F, const StoreLayout &S, int ky0, int kyloc0, int nky, const void *Y,
                           const void *twL, void *out, hipStream_t st) {
    constexpr int W = NC * FIELD_RB, threads = W * L / E;
    const size_t shmem = sizeof(double) * zdfft::ColsInner<L, W>::SIZE;
    set_dyn_lds<k_zfft_f<L, E, NC>>(shmem);
    if (nky % FIELD_RB || kyloc0 % FIELD_RB) return 2;
    dim3 grid(S.N / NC, nky / FIELD_RB, F.nfield), block(threads);
    hipLaunchKernelGGL((k_zfft_f<L, E, NC>), grid, block, shmem, st, F, S, ky0, kyloc0, nky, (const cplx *) Y,
                       (const cplx *) twL, (cplx *) out);
    ZD_LAUNCH_CHECK();
    return 0;
}
// columns per z-FFT workgroup of the field store (x 8 rows = lines per workgroup); the generator prunes by it
int zfft_fields_tile_columns(int L) {
    if (L & (L - 1)) return zfft_fields_np2_columns(L);  // composite lengths: zd_kernels_np2.hip
    switch (L) {
        case 32: case 64: case 128: return 4;
        case 256: return 2;
        case 512: case 1024: case 2048: return 1;  // 8 lines: 256 threads at L = 512 fit beside three generator workgroups
    }
    return 0;
}
int launch_zfft_fields(int L, const FieldLayout &F, const StoreLayout &S, int ky0, int kyloc0, int nky, const void *Y,
                       const void *twL, void *out, hipStream_t st) {
#define ZCASE(l, e, nc) \
    case l: return launch_zfft_f_t<l, e, nc>(F, S, ky0, kyloc0, nky, Y, twL, out, st);
#ifdef ZD_TUNING
    // A/B (VERDICT r4 #5 ii): the 512-point z FFT of the default workload with 8 elements per thread — 512 threads of half the
    // registers, two waves per SIMD beside the generator's three instead of one
    if (L == 512 && getenv("ZD_ZFFT_E8")) return launch_zfft_f_t<512, 8, 1>(F, S, ky0, kyloc0, nky, Y, twL, out, st);
#endif
    switch (L) {  // = zfft_fields_tile_columns
        ZCASE(32, 16, 4)
        ZCASE(64, 16, 4)
        ZCASE(128, 16, 4)
        ZCASE(256, 16, 2)
        ZCASE(512, 16, 1)
        ZCASE(1024, 16, 1)
        ZCASE(2048, 16, 1)
    }
#undef ZCASE
    fprintf(stderr, "zeldovich_hip: field store: unsupported z-FFT length %d (power of two in [32,2048] required)\n", L);
    return 2;
}

template <int N, int E, int W, bool PERSIST = false, int MINW = 1>
static int launch_yfft_f_t(const FieldLayout &F, const StoreLayout &S, const void *tw, const void *store, int plane0,
                           int nplanes, int ring_pitch, void *ring, hipStream_t st) {
    constexpr int threads = W * N / E;
    // the transform's exchange area + the row-block table (8 B per block of FIELD_RB row slots)
    const size_t shmem = sizeof(double) * (zdfft::ColsInner<N, W>::SIZE + (size_t) ((S.Hq + FIELD_RB - 1) / FIELD_RB));
    // (the attribute is set once per device: for the largest table, one rank's)
    const size_t shmem_max = std::min<size_t>(160 * 1024, sizeof(double) * (zdfft::ColsInner<N, W>::SIZE + (size_t) (N / 2 + FIELD_RB - 1) / FIELD_RB));
    if (shmem > 160 * 1024) {
        fprintf(stderr, "zeldovich_hip: y stage of the field store needs %zu B of LDS at PPD %d\n", shmem, N);
        return 2;
    }
    if constexpr (PERSIST) {
        set_dyn_lds<k_yfft_fp<N, E, W, MINW>>(shmem_max);
        int dev = 0, ncu = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const int per_cu = std::max(1, std::min((int) (160 * 1024 / shmem), 2048 / (threads * (MINW > 1 ? 1 : 2))));
        dim3 grid((unsigned) std::max(8, ncu * per_cu / 8 * 8)), block(threads);
        hipLaunchKernelGGL((k_yfft_fp<N, E, W, MINW>), grid, block, shmem, st, F, S, (const cplx *) tw, (const cplx *) store, plane0,
                           ring_pitch, (cplx *) ring, nplanes);
        ZD_LAUNCH_CHECK();
    } else {
        set_dyn_lds<k_yfft_f<N, E, W, MINW>>(shmem_max);
        dim3 grid(3 * (N / W), 1, nplanes), block(threads);
        hipLaunchKernelGGL((k_yfft_f<N, E, W, MINW>), grid, block, shmem, st, F, S, (const cplx *) tw, (const cplx *) store, plane0,
                           ring_pitch, (cplx *) ring);
    }
    ZD_LAUNCH_CHECK();
    return 0;
}
// y stage of the field store: store planes [plane0, plane0 + nplanes) -> ring planes [0, nplanes)
int launch_yfft_fields(const FieldLayout &F, const StoreLayout &S, const void *tw, const void *store, int plane0,
                       int nplanes, int ring_pitch, void *ring, hipStream_t st) {
#define YCASE(n, e, w) \
    case n: return launch_yfft_f_t<n, e, w>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
#ifdef ZD_TUNING
    {
        const int yw = getenv("ZD_YW") ? atoi(getenv("ZD_YW")) : 0, yp = getenv("ZD_YPERSIST") ? atoi(getenv("ZD_YPERSIST")) : 0;
        if (S.N == 4096 && yw == 2 && !yp) return launch_yfft_f_t<4096, 16, 2, false, 4>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
        if (S.N == 4096 && yw == 2 && yp) return launch_yfft_f_t<4096, 16, 2, true, 4>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
        if (S.N == 4096 && yp) return launch_yfft_f_t<4096, 16, 4, true>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
        if (S.N == 2048 && yw == 4 && !yp) return launch_yfft_f_t<2048, 16, 4, false, 4>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
        if (S.N == 2048 && yw == 4 && yp) return launch_yfft_f_t<2048, 16, 4, true, 4>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
        if (S.N == 2048 && yp) return launch_yfft_f_t<2048, 16, 8, true>(F, S, tw, store, plane0, nplanes, ring_pitch, ring, st);
    }
#endif
    switch (S.N) {
        YCASE(64, 16, 32)
        YCASE(128, 16, 32)
        YCASE(256, 16, 16)
        YCASE(512, 16, 16)
        YCASE(1024, 16, 8)
        YCASE(2048, 16, 8)
        YCASE(4096, 16, 4)
        YCASE(8192, 16, 2)
        YCASE(16384, 16, 1)
    }
#undef YCASE
    fprintf(stderr, "zeldovich_hip: field store unsupported for PPD %d\n", S.N);
    return 2;
}

int launch_yfft(const StoreLayout &S, int nplanes, const void *tw, void *data, hipStream_t st) {
#define YCASE(n, e, w) \
    case n: return launch_yfft_t<n, e, w>(S, nplanes, tw, data, st);
    switch (S.N) {
        YCASE(32, 16, 32)
        YCASE(64, 16, 32)
        YCASE(128, 16, 32)
        YCASE(256, 16, 16)
        YCASE(512, 16, 16)
        YCASE(1024, 16, 8)
        YCASE(2048, 16, 8)
        YCASE(4096, 16, 4)
        YCASE(8192, 16, 2)
    }
#undef YCASE
    fprintf(stderr, "zeldovich_hip: unsupported PPD %d (power of two in [32,8192] required)\n", S.N);
    return 2;
}

template <int N, int E, int NA, int ROWS>
static int launch_xfft_t(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0,
                         int nplanes, int z_first, int z_step, void *records, float *density, Reduce *red, hipStream_t st) {
    constexpr int WL = ROWS * NA, threads = WL * N / E;
    constexpr size_t fft_dbl = zdfft::LineInner<N, WL>::SIZE, fld_dbl = (size_t) ROWS * 2 * NA * N / XFFT_NH(N, NA, ROWS);
    // PACK_PLT3: lines 0 and 2 hand their results to line 1 through LDS, 2 x 16 B per column and row (k_xfft)
    const size_t plt_dbl = (NA == 3 && ec.pack == PACK_PLT3) ? (size_t) 4 * ROWS * N : 0;
    const size_t base_dbl = fft_dbl > fld_dbl ? fft_dbl : fld_dbl;
    const size_t shmem = sizeof(double) * (base_dbl > plt_dbl ? base_dbl : plt_dbl);
    if (threads > 1024) {
        fprintf(stderr, "zeldovich_hip: x pass for PPD %d with %d arrays needs %d threads per workgroup: unsupported\n", N, NA, threads);
        return 2;
    }
    if (shmem > 160 * 1024) {
        fprintf(stderr, "zeldovich_hip: x pass for PPD %d with %d arrays needs %zu B of LDS (> 160 KB): unsupported\n", N, NA, shmem);
        return 2;
    }
    dim3 grid(N / ROWS, nplanes), block(threads);
    if constexpr (NA == 3) {
        if (ec.pack == PACK_PLT3) {
            set_dyn_lds<k_xfft<N, E, NA, ROWS, true>>(shmem);
            hipLaunchKernelGGL((k_xfft<N, E, NA, ROWS, true>), grid, block, shmem, st, S, ec, (const cplx *) tw, (const cplx *) data, plane0,
                               z_first, z_step, (char *) records, density, red);
            ZD_LAUNCH_CHECK();
            return 0;
        }
    }
    set_dyn_lds<k_xfft<N, E, NA, ROWS>>(shmem);
    hipLaunchKernelGGL((k_xfft<N, E, NA, ROWS>), grid, block, shmem, st, S, ec, (const cplx *) tw, (const cplx *) data,
                       plane0, z_first, z_step, (char *) records, density, red);
    ZD_LAUNCH_CHECK();
    return 0;
}
template <int N, int E>
static int launch_xfft_seq_t(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0,
                             int nplanes, int z_first, int z_step, void *records, Reduce *red, hipStream_t st) {
    const size_t shmem = sizeof(double) * (zdfft::LineInner<N, 1>::SIZE + N);  // + Im of array 2 (k_xfft_seq)
    set_dyn_lds<k_xfft_seq<N, E>>(shmem);
    dim3 grid(N, nplanes), block(N / E);
    hipLaunchKernelGGL((k_xfft_seq<N, E>), grid, block, shmem, st, S, ec, (const cplx *) tw, (const cplx *) data, plane0, z_first,
                       z_step, (char *) records, red);
    ZD_LAUNCH_CHECK();
    return 0;
}

template <int N, int E, bool SPLIT2 = false>
static int launch_xfft_seq_plt_t(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0,
                                 int nplanes, int z_first, int z_step, void *records, Reduce *red, hipStream_t st) {
    const size_t shmem = sizeof(double) * (zdfft::LineInner<N, 1>::SIZE + (SPLIT2 ? 1 : 2) * N);  // + {vy, vz} (or vz) of array 2
    set_dyn_lds<k_xfft_seq_plt<N, E, SPLIT2>>(shmem);
    dim3 grid(N, nplanes), block(N / E);
    hipLaunchKernelGGL((k_xfft_seq_plt<N, E, SPLIT2>), grid, block, shmem, st, S, ec, (const cplx *) tw, (const cplx *) data, plane0, z_first,
                       z_step, (char *) records, red);
    ZD_LAUNCH_CHECK();
    return 0;
}

template <int N, int E>
static int launch_xfft_q2_plt_t(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0, int nplanes,
                                int z_first, int z_step, void *records, Reduce *red, hipStream_t st) {
#ifdef ZD_TUNING
    if (getenv("ZD_XQ_NP4")) {  // A/B: one workgroup per plane group (all four planes: whole 64-byte pieces per column, one workgroup per CU)
        const size_t sh4 = sizeof(double) * (zdfft::ColsInner<N, 4>::SIZE + 4 * N);
        set_dyn_lds<k_xfft_q2_plt<N, E, 4>>(sh4);
        const int q4 = ((plane0 + nplanes - 1) >> 2) - (plane0 >> 2) + 1;
        hipLaunchKernelGGL((k_xfft_q2_plt<N, E, 4>), dim3(N, q4), dim3(4 * N / E), sh4, st, S, ec, (const cplx *) tw, (const cplx *) data, plane0,
                           nplanes, z_first, z_step, (char *) records, red, 1);
        ZD_LAUNCH_CHECK();
        return 0;
    }
#endif
    const size_t shmem = sizeof(double) * (zdfft::ColsInner<N, 2>::SIZE + 2 * N);  // + vz of array 2, both lines
    set_dyn_lds<k_xfft_q2_plt<N, E>>(shmem);
    const int quads = ((plane0 + nplanes - 1) >> 2) - (plane0 >> 2) + 1;
    dim3 grid(2 * N, quads), block(2 * N / E);
    // distance, in positions of an XCD's dispatch stream, between the two workgroups that read the same lines (see the kernel)
    int hdist = 1;
#ifdef ZD_TUNING
    if (const char *env = getenv("ZD_XQ_DIST")) hdist = std::max(1, std::min(N / 8, atoi(env)));
#endif
    hipLaunchKernelGGL((k_xfft_q2_plt<N, E>), grid, block, shmem, st, S, ec, (const cplx *) tw, (const cplx *) data, plane0, nplanes,
                       z_first, z_step, (char *) records, red, hdist);
    ZD_LAUNCH_CHECK();
    return 0;
}
template <int N, int E, bool PLT>
static int launch_xfft_two_t(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0,
                             int nplanes, int z_first, int z_step, void *records, Reduce *red, hipStream_t st) {
    const size_t shmem = sizeof(double) * zdfft::LineInner<N, 1>::SIZE;
    set_dyn_lds<k_xfft_two<N, E, PLT>>(shmem);
    for (int emit = 0; emit < 2; emit++) {
        dim3 grid(N, nplanes, (emit != 0) == PLT ? 1 : 2), block(N / E);
        hipLaunchKernelGGL((k_xfft_two<N, E, PLT>), grid, block, shmem, st, S, ec, (const cplx *) tw, (cplx *) data, emit, plane0, z_first,
                           z_step, (char *) records, red);
        ZD_LAUNCH_CHECK();
    }
    return 0;
}

int launch_xfft(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0, int nplanes,
                int z_first, int z_step, void *records, float *density, Reduce *red, hipStream_t st) {
    if (S.lq) {  // plane-interleaved rows: the store of the fused PLT Z stage (zd_kernels_fz.hip)
        if (S.lq == 2 && S.one_block && S.narray == 3 && ec.pack == PACK_PLT3 && !density) {
            if (S.N == 2048) return launch_xfft_q2_plt_t<2048, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
            if (S.N == 1024) return launch_xfft_q2_plt_t<1024, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
            if (S.N == 512) return launch_xfft_q2_plt_t<512, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
        }
        fprintf(stderr, "zeldovich_hip: no x pass for plane-interleaved rows at PPD %d\n", S.N);
        return 2;
    }
    // PPD = 8192 with the field store's ring: three lines of a row are 1536 threads / 209 KB of LDS -> one row per
    // workgroup, its arrays in sequence
    if (S.N == 8192 && S.narray == 3 && ec.pack == PACK_ZAFIELD)
        return launch_xfft_seq_t<8192, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    // PPD = 4096 too (round 3): the one-row form is 256 threads, 190 registers and 68 KB of LDS — TWO independent workgroups per
    // CU, whose load / transform / store phases overlap — where the three-line k_xfft is one workgroup of 768 threads and 104 KB:
    // x stage 798 -> 713 ms (3.85 TB at 5.4 TB/s)
    // (8 elements per thread — 512 threads at 116 registers, twice the waves in the same two workgroups — is slower: 714 vs 699 ms)
    if (S.N == 4096 && S.narray == 3 && ec.pack == PACK_ZAFIELD)
        return launch_xfft_seq_t<4096, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    if (S.N == 1024 && S.narray == 3 && ec.pack == PACK_ZAFIELD)  // one wave per workgroup: 13.5 -> 11.0 ms
        return launch_xfft_seq_t<1024, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    if (S.N == 2048 && S.narray == 3 && ec.pack == PACK_ZAFIELD)  // 128 threads, four workgroups per CU: 101.6 -> 89.5 ms
        return launch_xfft_seq_t<2048, 16>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    if (S.N == 16384 && S.narray == 3 && ec.pack == PACK_ZAFIELD)
        return launch_xfft_two_t<16384, 16, false>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    if (S.N == 8192 && S.narray == 3 && ec.pack == PACK_PLT3)  // the ring of the PLT field store
        return launch_xfft_two_t<8192, 16, true>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    // PPD = 2048, PLT3 packing: the one-row form too (k_xfft_seq_plt: 128 threads, 204 registers, 49 KB of LDS: three workgroups
    // per CU).  x stage: two rows per k_xfft workgroup (768 threads) 164.6 ms, one row (384 threads, two workgroups per CU) 151.3,
    // one row per k_xfft_seq_plt workgroup 137.0.  (At PPD = 4096 the row's second array would need 64 KB of LDS beside the
    // transform's 35 — one workgroup of 256 threads per CU: 1444 ms against k_xfft's 1338 — hence the SPLIT2 form above.)
    // PPD = 4096: the same with vy kept in registers as well (240 registers, 67 KB of LDS: two workgroups of 256 threads per CU):
    // x stage of PPD=4096 PLT 1338 -> 1067 ms
    if (S.N == 4096 && S.narray == 3 && ec.pack == PACK_PLT3)
        return launch_xfft_seq_plt_t<4096, 16, true>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);

    // (with vy in registers as well — 240 registers, 33 KB of LDS, four workgroups per CU — 134.6 -> 127.8)
    if (S.N == 2048 && S.narray == 3 && ec.pack == PACK_PLT3)
        return launch_xfft_seq_plt_t<2048, 16, true>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, red, st);
    // (the ZA field ring at 2048 is indifferent: 104.2 ms with two rows, 105.7 with one)
#define XCASE(n, e, rows1, rows2, rows4, rows3z, rows3p)                                                              \
    case n:                                                                                                           \
        if (S.narray == 1) return launch_xfft_t<n, e, 1, rows1>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, density, red, st); \
        if (S.narray == 2) return launch_xfft_t<n, e, 2, rows2>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, density, red, st); \
        if (S.narray == 3 && ec.pack == PACK_ZAPAIR)                                                                  \
            return launch_xfft_t<n, e, 3, rows3z>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, density, red, st); \
        if (S.narray == 3) return launch_xfft_t<n, e, 3, rows3p>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, density, red, st); \
        return launch_xfft_t<n, e, 4, rows4>(S, ec, tw, data, plane0, nplanes, z_first, z_step, records, density, red, st);
    // rows per workgroup for 1 / 2 / 4 arrays and for the packed stores (ZA pairs / PLT), measured
    switch (S.N) {
        XCASE(32, 16, 32, 32, 16, 16, 16)
        XCASE(64, 16, 32, 32, 16, 16, 16)
        XCASE(128, 16, 32, 16, 8, 8, 8)
        XCASE(256, 16, 16, 8, 4, 4, 4)
        XCASE(512, 16, 8, 4, 2, 2, 2)
        XCASE(1024, 16, 4, 2, 1, 2, 2)
        XCASE(2048, 16, 2, 1, 1, 1, 2)
        XCASE(4096, 16, 1, 1, 1, 1, 1)
        XCASE(8192, 16, 1, 1, 1, 1, 1)
    }
#undef XCASE
    fprintf(stderr, "zeldovich_hip: unsupported PPD %d\n", S.N);
    return 2;
}

#ifdef ZD_TESTING
template <int N, int E, int W>
static int launch_test_fft_t(int kind, const void *tw, const void *in, void *out, long long lines, hipStream_t st) {
    constexpr int threads = W * N / E;
    if (lines % W) return 3;
    if (kind == 1) {
        const size_t shmem = sizeof(double) * zdfft::ColsInner<N, W>::SIZE;
        set_dyn_lds<k_test_fft_cols<N, E, W>>(shmem);
        hipLaunchKernelGGL((k_test_fft_cols<N, E, W>), dim3((unsigned) (lines / W)), dim3(threads), shmem, st,
                           (const cplx *) tw, (const cplx *) in, (cplx *) out, lines);
    } else {
        const size_t shmem = sizeof(double) * zdfft::LineInner<N, W>::SIZE;
        set_dyn_lds<k_test_fft_lines<N, E, W>>(shmem);
        hipLaunchKernelGGL((k_test_fft_lines<N, E, W>), dim3((unsigned) (lines / W)), dim3(threads), shmem, st,
                           (const cplx *) tw, (const cplx *) in, (cplx *) out, lines);
    }
    ZD_LAUNCH_CHECK();
    return 0;
}
int launch_test_fft(int n, int kind, const void *tw, const void *in, void *out, long long lines, hipStream_t st) {
#define TCASE(nn, e, w) \
    case nn: return launch_test_fft_t<nn, e, w>(kind, tw, in, out, lines, st);
    switch (n) {
        TCASE(32, 16, 32)
        TCASE(64, 16, 32)
        TCASE(128, 16, 32)
        TCASE(256, 16, 16)
        TCASE(512, 16, 16)
        TCASE(1024, 16, 8)
        TCASE(2048, 16, 8)
        TCASE(4096, 16, 4)
        TCASE(8192, 16, 2)
        TCASE(16384, 16, 1)
    }
#undef TCASE
    return 2;
}
int test_fft_tile_width(int n) { return zfft_tile_width(n); }

#endif  // ZD_TESTING
#ifdef ZD_TUNING
// ---- tuning harness: the y pass in alternative tile shapes (zd_test_yfft_variant) ----
template <int N, int E, int W, int MINW>
static int launch_yfft_v(const StoreLayout &S, int nplanes, const void *tw, void *data, hipStream_t st) {
    constexpr int threads = W * N / E;
    const size_t shmem = sizeof(double) * zdfft::ColsInner<N, W>::SIZE;
    set_dyn_lds<k_yfft<N, E, W, MINW>>(shmem);
    dim3 grid(N / W, S.narray, nplanes), block(threads);
    hipLaunchKernelGGL((k_yfft<N, E, W, MINW>), grid, block, shmem, st, S, (const cplx *) tw, (cplx *) data);
    ZD_LAUNCH_CHECK();
    return 0;
}
int launch_yfft_variant(int variant, const StoreLayout &S, int nplanes, const void *tw, void *data, hipStream_t st) {
#define VC(id, n, e, w, m) \
    if (S.N == n && variant == id) return launch_yfft_v<n, e, w, m>(S, nplanes, tw, data, st);
    VC(0, 2048, 16, 4, 1) VC(1, 2048, 16, 4, 4) VC(2, 2048, 16, 8, 4) VC(3, 2048, 8, 4, 4) VC(4, 2048, 8, 2, 4)
    VC(5, 2048, 16, 2, 4) VC(6, 2048, 8, 4, 8) VC(7, 2048, 16, 2, 2) VC(8, 2048, 4, 2, 8) VC(9, 2048, 8, 2, 8)
    VC(0, 1024, 16, 8, 1) VC(1, 1024, 16, 8, 4) VC(2, 1024, 16, 16, 4) VC(3, 1024, 8, 8, 4) VC(4, 1024, 8, 4, 4)
    VC(5, 1024, 16, 4, 4) VC(6, 1024, 8, 8, 8) VC(7, 1024, 16, 4, 2) VC(8, 1024, 4, 4, 8) VC(9, 1024, 8, 4, 8)
#undef VC
    return 2;
}

#endif  // ZD_TUNING

int launch_fnl_table(const GenConst &g, int n, void *tab, hipStream_t st) {
    dim3 grid((n + 255) / 256), block(256);
    if (g.is_powerlaw)
        hipLaunchKernelGGL(k_fnl_table<true>, grid, block, 0, st, g, n, (double *) tab);
    else
        hipLaunchKernelGGL(k_fnl_table<false>, grid, block, 0, st, g, n, (double *) tab);
    ZD_LAUNCH_CHECK();
    return 0;
}

template <int N, int E, int W, int ROWS>
static int launch_fnl_t(int which, const StoreLayout &S, double f_NL, const void *tw, void *data, void *phik, int nplanes, int lZq,
                        hipStream_t st) {
    if (which == 0) {  // x: inverse + nonlinearity + forward
        constexpr int threads = ROWS * N / E;
        const size_t shmem = sizeof(double) * zdfft::LineInner<N, ROWS>::SIZE;
        set_dyn_lds<k_xphi<N, E, ROWS>>(shmem);
        const double inv = 1. / N / N / N;
        hipLaunchKernelGGL((k_xphi<N, E, ROWS>), dim3(N / ROWS, nplanes), dim3(threads), shmem, st, S, f_NL, inv, (const cplx *) tw,
                           (cplx *) data);
        ZD_LAUNCH_CHECK();
    } else {
        constexpr int threads = W * N / E;
        const size_t shmem = sizeof(double) * zdfft::ColsInner<N, W>::SIZE;
        if (which == 1) {
            set_dyn_lds<k_yfwd<N, E, W>>(shmem);
            hipLaunchKernelGGL((k_yfwd<N, E, W>), dim3(N / W, 1, nplanes), dim3(threads), shmem, st, S, (const cplx *) tw, (cplx *) data);
            ZD_LAUNCH_CHECK();
        } else {
            set_dyn_lds<k_zfwd<N, E, W>>(shmem);
            hipLaunchKernelGGL((k_zfwd<N, E, W>), dim3(N / W, S.Hq), dim3(threads), shmem, st, S, lZq, (const cplx *) tw,
                               (const cplx *) data, (cplx *) phik);
            ZD_LAUNCH_CHECK();
        }
    }
    return 0;
}
// which: 0 = x pass (inverse, phi + f_NL phi^2, forward) and 1 = forward y, on the planes [0, nplanes) of `data` (a store or a
// ring slot of the exchange); 2 = forward z over this rank's S.Hq row slots and all N planes (lZq = log2 of the planes per
// chunk) -> phik
int launch_fnl_stage(int which, const StoreLayout &S, double f_NL, const void *tw, void *data, void *phik, int nplanes, int lZq,
                     hipStream_t st) {
#define FCASE(n, e, w, rows) \
    case n: return launch_fnl_t<n, e, w, rows>(which, S, f_NL, tw, data, phik, nplanes, lZq, st);
    switch (S.N) {
        FCASE(32, 16, 32, 32)
        FCASE(64, 16, 32, 32)
        FCASE(128, 16, 32, 32)
        FCASE(256, 16, 16, 16)
        FCASE(512, 16, 16, 8)
        FCASE(1024, 16, 8, 4)
        FCASE(2048, 16, 8, 2)
        FCASE(4096, 16, 4, 1)
    }
#undef FCASE
    fprintf(stderr, "zeldovich_hip: f_NL path supports PPD = 32..4096 (power of two), got %d\n", S.N);
    return 2;
}

#ifdef ZD_TUNING
int launch_copy16(const void *in, void *out, long long n16, hipStream_t st) {
    hipLaunchKernelGGL(k_copy16, dim3(256 * 8), dim3(256), 0, st, (const uint4 *) in, (uint4 *) out, n16);
    ZD_LAUNCH_CHECK();
    return 0;
}

#endif

}  // namespace zd
