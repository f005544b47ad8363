/* TEST INFRASTRUCTURE ONLY — see zd_oracle.h for the scope and the parity-pinning statement.
 *
 * Plain-C restatement of the reference's grid->displacements algorithm.  It deliberately keeps the
 * reference's structure (per-plane sequential RNG with skip bookkeeping, slab / reflected slab,
 * displaced twin storage, block array with the y-shift on load, Nyquist-row zeroing, 2-D FFT,
 * particle epilogue) so that it can be compared line by line with the cited sources.  The only
 * substitution is the FFT backend: the reference calls FFTW3 (absent from this image); a DFT is
 * uniquely defined, so an iterative radix-2 transform with long-double twiddles is used instead.
 */
#include "zd_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------- */
/* pcg64 = setseq_xsl_rr_128_64 (include/pcg-rng/pcg_random.hpp:1868,1965)                       */

/* default 128-bit multiplier and increment: pcg_random.hpp:159-170 */
#define PCG_MULT ((((u128) 0x2360ed051fc65da4ULL) << 64) | 0x4385df649fccf645ULL)
#define PCG_INC ((((u128) 0x5851f42d4c957f2dULL) << 64) | 0x14057b7ef767814fULL)

static inline u128 st_get(const zdo_pcg *g) { return (((u128) g->hi) << 64) | g->lo; }
static inline void st_set(zdo_pcg *g, u128 s) {
    g->hi = (uint64_t) (s >> 64);
    g->lo = (uint64_t) s;
}

/* engine(itype state): state_ = bump(state + increment())   pcg_random.hpp:427-432 */
void zdo_pcg_seed(zdo_pcg *g, uint64_t seed) {
    u128 s = (u128) seed + PCG_INC;
    s      = s * PCG_MULT + PCG_INC;
    st_set(g, s);
}

/* xsl_rr_mixin::output for 128->64: pcg_random.hpp:1144-1170 */
static inline uint64_t xsl_rr(u128 s) {
    uint64_t x   = (uint64_t) (s >> 64) ^ (uint64_t) s;
    unsigned rot = (unsigned) (s >> 122);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

/* operator(): for 128-bit state output_previous is false, i.e. advance THEN output
 * (pcg_random.hpp:381-386, 855) */
uint64_t zdo_pcg_next(zdo_pcg *g) {
    u128 s = st_get(g) * PCG_MULT + PCG_INC;
    st_set(g, s);
    return xsl_rr(s);
}

/* engine::advance(delta): square-and-multiply on (mult, plus)   pcg_random.hpp:657-687 */
static u128 lcg_advance(u128 state, u128 delta, u128 cur_mult, u128 cur_plus) {
    u128 acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}
void zdo_pcg_advance(zdo_pcg *g, uint64_t delta_hi, uint64_t delta_lo) {
    u128 d = (((u128) delta_hi) << 64) | delta_lo;
    st_set(g, lcg_advance(st_get(g), d, PCG_MULT, PCG_INC));
}

/* engine::distance (operator-): bit-by-bit reconstruction   pcg_random.hpp:732-764 */
uint64_t zdo_pcg_distance(const zdo_pcg *a, const zdo_pcg *b) {
    u128 cur_state = st_get(a), newstate = st_get(b);
    u128 cur_mult = PCG_MULT, cur_plus = PCG_INC;
    u128 the_bit = 1, distance = 0;
    while (cur_state != newstate) {
        if ((cur_state & the_bit) != (newstate & the_bit)) {
            cur_state = cur_state * cur_mult + cur_plus;
            distance |= the_bit;
        }
        the_bit <<= 1;
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
    }
    return (uint64_t) distance;
}

/* one_rand<2>: uint64 -> (0,1]   src/power_spectrum.cpp:284-308 */
double zdo_u01(uint64_t r) {
    if (r == UINT64_MAX) return 1.;
    r += (uint64_t) 1;
    return ldexp((double) r, -64);
}

/* ------------------------------------------------------------------------------------------- */
/* SplineFunction (include/spline_function.h)                                                  */

/* sort_arrays: shell sort of (x,y) by x   spline_function.h:77-104 */
static void spline_sort(int n, double *x, double *y) {
    int i, j, inc;
    double v, w;
    double *a = x - 1, *b = y - 1; /* 1-based views as in the source */
    inc       = 1;
    do {
        inc *= 3;
        inc++;
    } while (inc <= n);
    do {
        inc /= 3;
        for (i = inc + 1; i <= n; i++) {
            v = a[i];
            w = b[i];
            j = i;
            /* NB the reference tests the ZERO-based array here (x[j-inc], not a[j-inc]);
             * restated literally (spline_function.h:95) */
            while (x[j - inc] > v) {
                a[j] = a[j - inc];
                b[j] = b[j - inc];
                j -= inc;
                if (j <= inc) break;
            }
            a[j] = v;
            b[j] = w;
        }
    } while (inc > 1);
}

/* spline(): natural boundary conditions at both ends   spline_function.h:106-139 */
void zdo_spline_build(int n, double *x, double *y, double *y2) {
    int i, k;
    double p, qn, sig, un;
    double *u = (double *) malloc(sizeof(double) * (size_t) n);
    spline_sort(n, x, y);
    y2[0] = u[0] = 0.0;
    for (i = 1; i <= n - 2; i++) {
        sig   = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
        p     = sig * y2[i - 1] + 2.0;
        y2[i] = (sig - 1.0) / p;
        u[i]  = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
        u[i]  = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
    }
    qn = un   = 0.0;
    y2[n - 1] = (un - qn * u[n - 2]) / (qn * y2[n - 2] + 1.0);
    for (k = n - 2; k >= 0; k--) y2[k] = y2[k] * y2[k + 1] + u[k];
    free(u);
}

/* val(): bisection + cubic   spline_function.h:141-163 */
double zdo_spline_val(int n, const double *x, const double *y, const double *y2, double v) {
    int klo = 0, khi = n - 1, k;
    double h, b, a;
    while (khi - klo > 1) {
        k = (khi + klo) >> 1;
        if (x[k] > v)
            khi = k;
        else
            klo = k;
    }
    h = x[khi] - x[klo];
    a = (x[khi] - v) / h;
    b = (v - x[klo]) / h;
    return a * y[klo] + b * y[khi] + ((a * a * a - a) * y2[klo] + (b * b * b - b) * y2[khi]) * (h * h) / 6.0;
}

/* ------------------------------------------------------------------------------------------- */
/* PowerSpectrum (src/power_spectrum.cpp)                                                      */

/* power(): src/power_spectrum.cpp:225-261 */
double zdo_power(const zdo_pk *pk, double k) {
    if (k <= 0.0) return 0.0;
    if (pk->is_powerlaw) {
        return pow(k, pk->powerlaw_index) * exp(-k * k * pk->Pk_smooth2) * pk->normalization;
    }
    return exp(zdo_spline_val(pk->n, pk->x, pk->y, pk->y2, log(k)) - k * k * pk->Pk_smooth2)
           * pk->normalization;
}

/* sigmaR_integrand: src/power_spectrum.cpp:50-58 */
static double sigmaR_integrand(const zdo_pk *pk, double k) {
    double x = k * pk->Rnorm;
    double w;
    if (x <= 1e-3)
        w = 1 - x * x / 10.0;
    else
        w = 3.0 * (sin(x) - x * cos(x)) / x / x / x;
    return 0.5 / M_PI / M_PI * k * k * w * w * zdo_power(pk, k);
}

/* Romberg: src/power_spectrum.cpp:93-128.  The reference reduces the midpoint sum under OpenMP
 * (order thread-dependent at 1e-16); here it is summed serially in k order. */
#define ROMB_MAXITER 32
static double romberg(const zdo_pk *pk, double a, double b, double prec, double *obtprec) {
    int jj;
    double h, s, fourtokm1;
    static double TT[ROMB_MAXITER + 1][ROMB_MAXITER + 1];
    h        = 0.5 * (b - a);
    TT[0][1] = h * (sigmaR_integrand(pk, a) + sigmaR_integrand(pk, b));
    jj       = 0;
    do {
        jj++;
        s = 0;
        for (uint64_t k = 1; k <= (1ULL << (jj - 1)); k++) s += sigmaR_integrand(pk, a + (2 * k - 1) * h);
        TT[jj][1] = 0.5 * TT[jj - 1][1] + h * s;
        fourtokm1 = 1;
        for (int k = 2; k <= jj; k++) {
            fourtokm1 *= 4;
            TT[jj][k] = TT[jj][k - 1] + (TT[jj][k - 1] - TT[jj - 1][k - 1]) / (fourtokm1 - 1);
        }
        h *= 0.5;
        if (jj > 1 && fabs(TT[jj][jj] - TT[jj - 1][jj - 1]) < prec * fabs(TT[jj][jj])) break;
    } while (jj < ROMB_MAXITER);
    *obtprec = (TT[jj][jj] - TT[jj - 1][jj - 1]) / TT[jj][jj];
    return TT[jj][jj];
}

/* sigmaR: src/power_spectrum.cpp:60-89 */
double zdo_sigmaR(zdo_pk *pk, double R) {
    if (!pk->is_powerlaw) {
        double target_prec = 1e-6, precision = 1.0;
        pk->Rnorm     = R;
        double retval = sqrt(romberg(pk, 0, 10.0, target_prec, &precision));
        if (precision > target_prec) {
            fprintf(stderr, "zd_oracle: Romberg precision %g > target %g\n", precision, target_prec);
            exit(1);
        }
        return retval;
    } else {
        double n      = pk->powerlaw_index;
        double retval = 9 * pow(R, -n - 3) / (2 * M_PI * sqrt(M_PI)) * tgamma((3 + n) / 2.)
                        / (tgamma((2 - n) / 2.) * (n - 3) * (n - 1));
        return sqrt(retval * pk->normalization);
    }
}

/* Normalize: src/power_spectrum.cpp:186-223 (primordial_norm only feeds f_NL: out of scope) */
static void pk_normalize(zdo_pk *pk, double Pk_norm, double Pk_sigma, double Pk_sigma_ratio,
                         double Pk_smooth, int fix_to_mean, double boxsize) {
    pk->Pk_smooth2    = 0.0;
    pk->normalization = 1.0;
    if (Pk_norm > 0.0) {
        if (Pk_sigma > 0) {
            pk->normalization = Pk_sigma / zdo_sigmaR(pk, Pk_norm);
            pk->normalization *= pk->normalization;
        } else if (Pk_sigma_ratio > 0) {
            pk->normalization = Pk_sigma_ratio * Pk_sigma_ratio;
        } else {
            assert(Pk_sigma > 0 || Pk_sigma_ratio > 0);
        }
    }
    pk->normalization /= boxsize * boxsize * boxsize;
    pk->Pk_smooth2  = Pk_smooth * Pk_smooth;
    pk->fixed_power = fix_to_mean;
}

/* the node-loading loop of InitFromFile: src/power_spectrum.cpp:151-167 */
int zdo_pk_from_table(zdo_pk *pk, int n, const double *kk, const double *PP, double Pk_scale,
                      double Pk_norm, double Pk_sigma, double Pk_sigma_ratio, double Pk_smooth,
                      int fix_to_mean, double boxsize) {
    memset(pk, 0, sizeof(*pk));
    pk->x    = (double *) malloc(sizeof(double) * (size_t) (n + 1));
    pk->y    = (double *) malloc(sizeof(double) * (size_t) (n + 1));
    pk->y2   = (double *) malloc(sizeof(double) * (size_t) (n + 1));
    pk->kmin = 1.7976931348623157e308;
    pk->kmax = 2.2250738585072014e-308; /* numeric_limits<double>::min(), power_spectrum.cpp:9-10 */
    pk->powerlaw_index = 1000;
    int nn             = 0;
    for (int i = 0; i < n; i++) {
        double k = kk[i], P = PP[i];
        if (k < 0.0) continue;
        if (P < 0.0) continue;
        k *= Pk_scale;
        if (k > 0.0) {
            pk->x[nn] = log(k);
            pk->y[nn] = log(P);
            if (k < pk->kmin) pk->kmin = k;
        } else {
            pk->x[nn] = -1e3;
            pk->y[nn] = log(P);
        }
        if (k > pk->kmax) pk->kmax = k;
        nn++;
    }
    pk->n = nn;
    zdo_spline_build(nn, pk->x, pk->y, pk->y2);
    pk_normalize(pk, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth, fix_to_mean, boxsize);
    return 0;
}

/* InitFromFile: src/power_spectrum.cpp:130-171 */
int zdo_pk_from_file(zdo_pk *pk, const char *path, double Pk_scale, double Pk_norm, double Pk_sigma,
                     double Pk_sigma_ratio, double Pk_smooth, int fix_to_mean, double boxsize) {
    char line[200];
    FILE *fp = fopen(path, "r");
    if (!fp) {
        fprintf(stderr, "zd_oracle: power spectrum file \"%s\" not found\n", path);
        return 1;
    }
    int cap = 16384, n = 0;
    double *k = (double *) malloc(sizeof(double) * (size_t) cap);
    double *P = (double *) malloc(sizeof(double) * (size_t) cap);
    double kv = 0, Pv = 0; /* the reference does not reset these between lines either */
    while (fgets(line, 200, fp) != NULL) {
        if (line[0] == '#') continue;
        sscanf(line, "%lf %lf", &kv, &Pv);
        if (n < cap) {
            k[n] = kv;
            P[n] = Pv;
            n++;
        }
    }
    fclose(fp);
    int rc = zdo_pk_from_table(pk, n, k, P, Pk_scale, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth,
                               fix_to_mean, boxsize);
    free(k);
    free(P);
    return rc;
}

/* InitFromPowerLaw: src/power_spectrum.cpp:173-184 */
int zdo_pk_from_powerlaw(zdo_pk *pk, double index, double Pk_norm, double Pk_sigma,
                         double Pk_sigma_ratio, double Pk_smooth, int fix_to_mean, double boxsize) {
    memset(pk, 0, sizeof(*pk));
    pk->powerlaw_index = index;
    pk->is_powerlaw    = 1;
    pk->kmin           = 1e-4;
    pk->kmax           = 2.2250738585072014e-308;
    pk_normalize(pk, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth, fix_to_mean, boxsize);
    return 0;
}

void zdo_pk_free(zdo_pk *pk) {
    free(pk->x);
    free(pk->y);
    free(pk->y2);
    pk->x = pk->y = pk->y2 = NULL;
}

/* primordial_power / infer_Tk: src/power_spectrum.cpp:263-274 */
static double primordial_power(const zdo_pk *pk, double k) {
    if (k <= 0.0) return 0.0;
    return pk->primordial_norm * exp(log(k) * pk->n_s);
}
double zdo_infer_Tk(const zdo_pk *pk, double k) {
    if (k <= 0.0) return 1.0;
    return sqrt(zdo_power(pk, k) / primordial_power(pk, k));
}
void zdo_pk_set_primordial(zdo_pk *pk, double n_s) { /* Normalize: src/power_spectrum.cpp:221-222 */
    pk->n_s             = n_s;
    pk->primordial_norm = 1.;
    pk->primordial_norm = zdo_power(pk, pk->kmin) / primordial_power(pk, pk->kmin);
}

/* cgauss<2>: src/power_spectrum.cpp:338-359 */
static void cgauss2(const zdo_pk *pk, double wavenumber, zdo_pcg *rng, double out[2]) {
    double Pk    = zdo_power(pk, wavenumber);
    double R     = zdo_u01(zdo_pcg_next(rng));
    double theta = zdo_u01(zdo_pcg_next(rng));
    if (!pk->fixed_power)
        R = sqrt(-Pk * log(R));
    else
        R = sqrt(Pk);
    theta  = 2 * M_PI * theta;
    out[0] = R * cos(theta);
    out[1] = R * sin(theta);
}

/* gsl_rng_mt19937 (see zd_oracle.h): mt_set / mt_get / mt_get_double of gsl rng/mt.c */
void zdo_mt_seed(zdo_mt *g, unsigned long s) {
    if (s == 0) s = 4357;
    g->mt[0] = (uint32_t) (s & 0xffffffffUL);
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253U * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t) i;
    g->mti = 624;
}
uint32_t zdo_mt_next(zdo_mt *g) {
    uint32_t *mt = g->mt;
    if (g->mti >= 624) {
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk]     = mt[kk + 397] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfU : 0);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk]     = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfU : 0);
        }
        uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[623]    = mt[396] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfU : 0);
        g->mti     = 0;
    }
    uint32_t k = mt[g->mti++];
    k ^= k >> 11;
    k ^= (k << 7) & 0x9d2c5680U;
    k ^= (k << 15) & 0xefc60000U;
    k ^= k >> 18;
    return k;
}
double zdo_mt_uniform(zdo_mt *g) { return zdo_mt_next(g) / 4294967296.0; }

/* cgauss<1>: src/power_spectrum.cpp:310-332 (rejection Box-Muller, a variable number of draws per mode) */
static void cgauss1(const zdo_pk *pk, double wavenumber, zdo_mt *rng, double out[2]) {
    double Pk = zdo_power(pk, wavenumber);
    double phase1, phase2, r2;
    do {
        phase1 = zdo_mt_uniform(rng) * 2.0 - 1.0;
        phase2 = zdo_mt_uniform(rng) * 2.0 - 1.0;
        r2     = phase1 * phase1 + phase2 * phase2;
    } while (!(r2 < 1.0 && r2 > 0.0));
    if (pk->fixed_power)
        r2 = sqrt(Pk / r2);
    else
        r2 = sqrt(-Pk * log(r2) / r2);
    out[0] = phase1 * r2;
    out[1] = phase2 * r2;
}

/* ------------------------------------------------------------------------------------------- */
/* PLT eigenmodes (src/zeldovich.cpp:149-276)                                                  */

static void interp_eigmode(const double *eig_vecs, int64_t eig_vecs_ppd, int ikx, int iky, int ikz,
                           int64_t ppd, double *e) {
    int64_t halfppd = eig_vecs_ppd / 2 + 1;
    int64_t ppdhalf = eig_vecs_ppd / 2;
#define EIGMODE(_kx, _ky, _kz, _i) \
    (eig_vecs[(int64_t) (_kx) * eig_vecs_ppd * halfppd * 4 + (_ky) * halfppd * 4 + (_kz) * 4 + (_i)])
    if (eig_vecs_ppd % ppd == 0) { /* :161-170 exact stride */
        for (int i = 0; i < 4; i++)
            e[i] = EIGMODE(ikx * eig_vecs_ppd / ppd, iky * eig_vecs_ppd / ppd, ikz * eig_vecs_ppd / ppd, i);
        return;
    }
    double fx = ((double) eig_vecs_ppd) / ppd * ikx;
    double fy = ((double) eig_vecs_ppd) / ppd * iky;
    double fz = ((double) eig_vecs_ppd) / ppd * ikz;
    /* never interpolate across the +/- Nyquist seam: map upwards  (:176-183) */
    if (fx > ppdhalf && fx < halfppd) fx = floor(fx + 1);
    if (fy > ppdhalf && fy < halfppd) fy = floor(fy + 1);
    if (fz > ppdhalf && fz < halfppd) fz = floor(fz + 1);
    int ikx_l = (int) fx, ikx_h = ikx_l + 1;
    int iky_l = (int) fy, iky_h = iky_l + 1;
    int ikz_l = (int) fz, ikz_h = ikz_l + 1;
    if (ikx_h == eig_vecs_ppd) ikx_h = 0; /* wrap: :194-198 */
    if (iky_h == eig_vecs_ppd) iky_h = 0;
    if (ikz_h == eig_vecs_ppd) ikz_h = 0;
    fx -= ikx_l;
    fy -= iky_l;
    fz -= ikz_l;
    double f[8];
    f[0] = (1 - fx) * (1 - fy) * (1 - fz);
    f[1] = (1 - fx) * (1 - fy) * (fz);
    f[2] = (1 - fx) * (fy) * (1 - fz);
    f[3] = (1 - fx) * (fy) * (fz);
    f[4] = (fx) * (1 - fy) * (1 - fz);
    f[5] = (fx) * (1 - fy) * (fz);
    f[6] = (fx) * (fy) * (1 - fz);
    f[7] = (fx) * (fy) * (fz);
    for (int i = 0; i < 4; i++) {
        /* f == 0 corners are still read by the reference; when a high index falls outside the
         * half-space kz table (ikz_h == halfppd) the reference reads the next row; mirror that
         * only when the weight is non-zero to stay inside the allocation */
        double acc = 0;
        acc += f[0] * EIGMODE(ikx_l, iky_l, ikz_l, i);
        acc += f[1] * (f[1] != 0 ? EIGMODE(ikx_l, iky_l, ikz_h, i) : 0.0);
        acc += f[2] * (f[2] != 0 ? EIGMODE(ikx_l, iky_h, ikz_l, i) : 0.0);
        acc += f[3] * (f[3] != 0 ? EIGMODE(ikx_l, iky_h, ikz_h, i) : 0.0);
        acc += f[4] * (f[4] != 0 ? EIGMODE(ikx_h, iky_l, ikz_l, i) : 0.0);
        acc += f[5] * (f[5] != 0 ? EIGMODE(ikx_h, iky_l, ikz_h, i) : 0.0);
        acc += f[6] * (f[6] != 0 ? EIGMODE(ikx_h, iky_h, ikz_l, i) : 0.0);
        acc += f[7] * (f[7] != 0 ? EIGMODE(ikx_h, iky_h, ikz_h, i) : 0.0);
        e[i] = acc;
    }
#undef EIGMODE
}

/* get_eigenmode: src/zeldovich.cpp:229-276 */
void zdo_get_eigenmode(const double *eig, int64_t eig_ppd, int kx, int ky, int kz, int64_t ppd,
                       int qPLT, double out[4]) {
    if (qPLT) {
        int ikx   = kx < 0 ? ppd + kx : kx;
        int iky   = ky < 0 ? ppd + ky : ky;
        int ikz   = kz < 0 ? ppd + kz : kz;
        ikz       = ikz > ppd / 2 ? ppd - ikz : ikz;
        double k2 = kx * kx + ky * ky + kz * kz;
        double ehat[4];
        interp_eigmode(eig, eig_ppd, ikx, iky, ikz, ppd, ehat);
        ehat[2] *= copysign(1, kz);
        double ehatmag = sqrt(ehat[0] * ehat[0] + ehat[1] * ehat[1] + ehat[2] * ehat[2]);
        ehat[0] /= ehatmag;
        ehat[1] /= ehatmag;
        ehat[2] /= ehatmag;
        double norm = k2 / (kx * ehat[0] + ky * ehat[1] + kz * ehat[2]);
        if (k2 == 0.0 || !isfinite(norm)) norm = 0.0;
        out[0] = norm * ehat[0];
        out[1] = norm * ehat[1];
        out[2] = norm * ehat[2];
        out[3] = ehat[3];
    } else {
        out[0] = kx;
        out[1] = ky;
        out[2] = kz;
        out[3] = 1;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* FFT backend (replaces the FFTW3 calls of src/zeldovich.cpp:83-114): unnormalised, sign +1      */

typedef struct {
    int n;
    int log2n;      /* -1 if n is not a power of two */
    double *wr, *wi; /* exp(+2 pi i k / n), k < n */
    int *rev;
    void *fw_line;   /* FFTW3 (when the host has libfftw3.so.3): one length-n transform, arbitrary stride at execution */
} fft_plan;

/* FFTW3 is what the reference calls (src/zeldovich.cpp:39-135: fftw_plan_many_dft / fftw_plan_dft_2d, sign +1, wisdom).  It is
 * not part of this image; where a host has libfftw3.so.3 the oracle binds it at run time (dlopen, no link dependency) and uses
 * it for every 1-D pass — plans are made in fft_plan_create, outside the timed intervals, as the reference's wisdom file would —
 * after checking it against the built-in radix-2 transform on a random vector.  zdo_fft_backend() says which one runs. */
#include <dlfcn.h>
typedef void *(*fftw_plan_many_dft_t)(int, const int *, int, double *, const int *, int, int, double *, const int *, int, int, int,
                                      unsigned);
typedef void (*fftw_execute_dft_t)(void *, double *, double *);
typedef void (*fftw_destroy_plan_t)(void *);
typedef void *(*fftw_malloc_t)(size_t);
typedef void (*fftw_free_t)(void *);
static struct {
    int probed, ok;
    fftw_plan_many_dft_t plan_many;
    fftw_execute_dft_t exec;
    fftw_destroy_plan_t destroy;
    fftw_malloc_t fmalloc;
    fftw_free_t ffree;
} g_fftw;
static int g_fftw_off = 0; /* zdo_fft_use_fftw(0): force the built-in transform */
static void fftw_probe(void) {
    if (g_fftw.probed) return;
    g_fftw.probed = 1;
    void *h = dlopen("libfftw3.so.3", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    g_fftw.plan_many = (fftw_plan_many_dft_t) dlsym(h, "fftw_plan_many_dft");
    g_fftw.exec      = (fftw_execute_dft_t) dlsym(h, "fftw_execute_dft");
    g_fftw.destroy   = (fftw_destroy_plan_t) dlsym(h, "fftw_destroy_plan");
    g_fftw.fmalloc   = (fftw_malloc_t) dlsym(h, "fftw_malloc");
    g_fftw.ffree     = (fftw_free_t) dlsym(h, "fftw_free");
    g_fftw.ok = g_fftw.plan_many && g_fftw.exec && g_fftw.destroy && g_fftw.fmalloc && g_fftw.ffree;
}
const char *zdo_fft_backend(void) {
    fftw_probe();
    return (g_fftw.ok && !g_fftw_off) ? "fftw3" : "radix-2";
}
void zdo_fft_use_fftw(int on) { g_fftw_off = !on; }

static void fft_exec(const fft_plan *p, double *data, int64_t stride, double *tmp);
static fft_plan *fft_plan_create(int n) {
    fft_plan *p = (fft_plan *) calloc(1, sizeof(fft_plan));
    p->n        = n;
    p->wr       = (double *) malloc(sizeof(double) * (size_t) n);
    p->wi       = (double *) malloc(sizeof(double) * (size_t) n);
    for (int k = 0; k < n; k++) {
        long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double) k / (long double) n;
        p->wr[k]        = (double) cosl(ang);
        p->wi[k]        = (double) sinl(ang);
    }
    int l = 0;
    while ((1 << l) < n) l++;
    p->log2n = ((1 << l) == n) ? l : -1;
    if (p->log2n >= 0) {
        p->rev = (int *) malloc(sizeof(int) * (size_t) n);
        for (int i = 0; i < n; i++) {
            int r = 0;
            for (int b = 0; b < l; b++)
                if (i & (1 << b)) r |= 1 << (l - 1 - b);
            p->rev[i] = r;
        }
    }
    fftw_probe();
    if (g_fftw.ok && !g_fftw_off && n >= 2) {
        /* contiguous in-place plan of ONE line, FFTW_BACKWARD = +1, FFTW_MEASURE on scratch (aligned like the run's buffers are
         * not guaranteed to be: FFTW_UNALIGNED); strided lines are gathered into a contiguous tmp by fft_exec */
        double *scr = (double *) g_fftw.fmalloc(sizeof(double) * 2 * (size_t) n);
        const int nn = n;
        p->fw_line = g_fftw.plan_many(1, &nn, 1, scr, NULL, 1, n, scr, NULL, 1, n, +1, (0U) | (1U << 1) /* MEASURE | UNALIGNED */);
        if (p->fw_line) { /* self-check against the built-in transform */
            double *a = (double *) malloc(sizeof(double) * 4 * (size_t) n), *b = a + 2 * n;
            unsigned long long sd = 88172645463325252ULL;
            for (int i = 0; i < 2 * n; i++) {
                sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17;
                a[i] = b[i] = (double) (sd >> 11) / 9007199254740992.0 - 0.5;
            }
            g_fftw.exec(p->fw_line, a, a);
            void *keep = p->fw_line;
            p->fw_line = NULL;
            double *tmp = (double *) malloc(sizeof(double) * 2 * (size_t) n);
            fft_exec(p, b, 1, tmp);
            free(tmp);
            double err = 0, mx = 0;
            for (int i = 0; i < 2 * n; i++) {
                err = fmax(err, fabs(a[i] - b[i]));
                mx  = fmax(mx, fabs(b[i]));
            }
            free(a);
            if (err <= 1e-12 * mx * (p->log2n >= 0 ? p->log2n + 1 : 8)) p->fw_line = keep;
            else g_fftw.destroy(keep);
        }
        g_fftw.ffree(scr);
    }
    return p;
}
static void fft_plan_destroy(fft_plan *p) {
    if (p->fw_line) g_fftw.destroy(p->fw_line);
    free(p->wr);
    free(p->wi);
    free(p->rev);
    free(p);
}

/* in-place length-n transform of interleaved complex data with element stride `stride` (in
 * complex elements); tmp must hold 2*n doubles */
static void fft_exec(const fft_plan *p, double *data, int64_t stride, double *tmp) {
    int n = p->n;
    if (p->fw_line) { /* FFTW3: contiguous lines in place, strided ones through tmp */
        if (stride == 1) {
            g_fftw.exec(p->fw_line, data, data);
            return;
        }
        for (int i = 0; i < n; i++) {
            tmp[2 * i]     = data[2 * i * stride];
            tmp[2 * i + 1] = data[2 * i * stride + 1];
        }
        g_fftw.exec(p->fw_line, tmp, tmp);
        for (int i = 0; i < n; i++) {
            data[2 * i * stride]     = tmp[2 * i];
            data[2 * i * stride + 1] = tmp[2 * i + 1];
        }
        return;
    }
    if (p->log2n < 0) {
        /* naive DFT for non power-of-two lengths (tiny test cases only) */
        for (int j = 0; j < n; j++) {
            long double sr = 0, si = 0;
            for (int k = 0; k < n; k++) {
                int idx   = (int) (((int64_t) j * k) % n);
                double xr = data[2 * k * stride], xi = data[2 * k * stride + 1];
                sr += (long double) xr * p->wr[idx] - (long double) xi * p->wi[idx];
                si += (long double) xr * p->wi[idx] + (long double) xi * p->wr[idx];
            }
            tmp[2 * j]     = (double) sr;
            tmp[2 * j + 1] = (double) si;
        }
        for (int j = 0; j < n; j++) {
            data[2 * j * stride]     = tmp[2 * j];
            data[2 * j * stride + 1] = tmp[2 * j + 1];
        }
        return;
    }
    for (int i = 0; i < n; i++) {
        int r          = p->rev[i];
        tmp[2 * r]     = data[2 * i * stride];
        tmp[2 * r + 1] = data[2 * i * stride + 1];
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len) {
            for (int j = 0; j < half; j++) {
                double wr = p->wr[j * step], wi = p->wi[j * step];
                double *a = tmp + 2 * (i + j), *b = tmp + 2 * (i + j + half);
                double tr = b[0] * wr - b[1] * wi, ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr;
                b[1] = a[1] - ti;
                a[0] += tr;
                a[1] += ti;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        data[2 * i * stride]     = tmp[2 * i];
        data[2 * i * stride + 1] = tmp[2 * i + 1];
    }
}

/* forward (sign -1) transform = conj(inverse(conj x)); ForwardFFT_Yonly / Forward2dFFT (:116-135) */
static void conj_plane(double *p, int64_t count) {
    for (int64_t i = 0; i < count; i++) p[2 * i + 1] = -p[2 * i + 1];
}

/* InverseFFT_Yonly: 1-D transform along the first (long-stride) index of p[n][n]   :93-114 */
static void inverse_fft_first_index(const fft_plan *pl, double *p, int n) {
    double *tmp = (double *) malloc(sizeof(double) * 2 * (size_t) n);
    for (int j = 0; j < n; j++) fft_exec(pl, p + 2 * j, n, tmp);
    free(tmp);
}
/* Inverse2dFFT: n x n, in place, row-major   :88-92 */
static void inverse_fft_2d(const fft_plan *pl, double *p, int n) {
    double *tmp = (double *) malloc(sizeof(double) * 2 * (size_t) n);
    for (int r = 0; r < n; r++) fft_exec(pl, p + 2 * (int64_t) r * n, 1, tmp);
    for (int c = 0; c < n; c++) fft_exec(pl, p + 2 * c, n, tmp);
    free(tmp);
}

static void forward_fft_first_index(const fft_plan *pl, double *p, int n) {
    conj_plane(p, (int64_t) n * n);
    inverse_fft_first_index(pl, p, n);
    conj_plane(p, (int64_t) n * n);
}
static void forward_fft_2d(const fft_plan *pl, double *p, int n) {
    conj_plane(p, (int64_t) n * n);
    inverse_fft_2d(pl, p, n);
    conj_plane(p, (int64_t) n * n);
}

/* ------------------------------------------------------------------------------------------- */

int zdo_record_size(int icformat) {
    switch (icformat) {
        case 0: return 32; /* ZelParticle: 3 u16 + pad, 3 f64 */
        case 1: return 32; /* RVZelParticle: 3 u16 + pad, 6 f32 */
        case 2: return 56; /* RVdoubleZelParticle: 3 u16 + pad, 6 f64 */
        case 3: return 12; /* ZelSimpleParticle: 3 f32 */
    }
    return -1;
}
int zdo_narray(const zdo_params *p) {
    if (p->qdensity == 2) return 1;
    return p->qPLT ? 4 : 2;
}

/* per-plane generator states: src/power_spectrum.cpp:26-37 */
static zdo_pcg *make_v2rng(const zdo_params *p) {
    int64_t nh   = p->ppd / 2;
    zdo_pcg *rng = (zdo_pcg *) malloc(sizeof(zdo_pcg) * (size_t) (nh > 0 ? nh : 1));
    unsigned long longseed = (unsigned long) (long) p->seed; /* :14 int -> unsigned long */
    zdo_pcg_seed(&rng[0], (uint64_t) longseed);
    for (int64_t i = 1; i < nh; i++) {
        rng[i] = rng[i - 1];
        zdo_pcg_advance(&rng[i], 0, (uint64_t) (2 * ZDO_MAX_PPD * ZDO_MAX_PPD));
    }
    return rng;
}

/* one mt19937 per yres, seeds seed + i: src/power_spectrum.cpp:18-25 */
static zdo_mt *make_v1rng(const zdo_params *p) {
    if (p->version != 1) return NULL;
    int64_t block = p->ppd / p->numblock;
    zdo_mt *rng   = (zdo_mt *) malloc(sizeof(zdo_mt) * (size_t) (block > 0 ? block : 1));
    unsigned long longseed = (unsigned long) (long) p->seed;
    for (int64_t i = 0; i < block; i++) zdo_mt_seed(&rng[i], longseed + (unsigned long) i);
    return rng;
}

typedef struct {
    int64_t ppd, ppdhalf, narray;
    int numblock, block;
} geom;

#define AYZX(_slab, _a, _y, _z, _x) \
    ((_slab) + 2 * ((int64_t) (_x) + g->ppd * ((_z) + g->ppd * ((_a) + g->narray * (int64_t) (_y)))))
#define AZYX(_slab, _a, _z, _y, _x) \
    ((_slab) + 2 * ((int64_t) (_x) + g->ppd * ((_y) + g->ppd * ((_a) + g->narray * (int64_t) (_z)))))

static inline void cset(double *d, double re, double im) {
    d[0] = re;
    d[1] = im;
}

/* LoadPlane without the trailing z FFTs: src/zeldovich.cpp:278-503 */
#define AYZX_PHI(_slab, _a, _y, _z, _x) ((_slab) + 2 * ((int64_t) (_x) + g->ppd * ((_z) + g->ppd * ((_a) + (int64_t) (_y)))))

static void load_plane_modes(const geom *g, const zdo_params *param, const zdo_pk *Pk, zdo_pcg *v2rng, zdo_mt *v1rng,
                             const double *eig, int64_t eig_ppd, int yblock, int yres, double *slab,
                             double *slabHer, int gen_phi, const double *input_phi_slab) {
    int64_t ppd = g->ppd, ppdhalf = g->ppdhalf;
    double fundamental2 = param->fundamental * param->fundamental;
    double ik_cutoff    = 1.0 / param->k_cutoff;
    int just_density    = param->qdensity == 2;
    double target_f     = (sqrt(1. + 24 * param->f_cluster) - 1) / 4.;
    double a_NL, a0;
    if (param->qPLTrescale) {
        a_NL = 1. / (1 + param->PLT_target_z);
        a0   = 1. / (1 + param->z_initial);
    } else {
        a_NL = a0 = 1.0;
    }
    int64_t nskip    = 0;
    double k2_cutoff = param->nyquist * param->nyquist / (param->k_cutoff * param->k_cutoff);
    int x, y, z, kx, ky, kz, xHer, yresHer, zHer;

    y                  = yres + yblock * g->block;
    zdo_pcg checkpoint = v2rng[y];
    ky                 = y > ppdhalf ? y - ppd : y;
    yresHer            = g->block - 1 - yres;
    for (z = 0; z < ppd; z++) {
        if (z == ppdhalf + 1) nskip += (ZDO_MAX_PPD - ppd) * ZDO_MAX_PPD;
        kz   = z > ppdhalf ? z - ppd : z;
        zHer = ppd - z;
        if (z == 0) zHer = 0;
        for (x = 0; x < ppd; x++) {
            if (x == ppdhalf + 1) nskip += ZDO_MAX_PPD - ppd;
            kx   = x > ppdhalf ? x - ppd : x;
            xHer = ppd - x;
            if (x == 0) xHer = 0;
            double k2   = (kx * kx + ky * ky + kz * kz) * fundamental2;
            double kmag = sqrt(k2);
            double D[2], F[2], G[2], H[2], f;
            int kmax = (double) ppdhalf * ik_cutoff + .5;
            if ((abs(kx) == kmax || abs(kz) == kmax || abs(ky) == kmax)
                || (!param->CornerModes && k2 >= k2_cutoff)
                || (param->qonemode
                    && !(kx == param->one_mode[0] && ky == param->one_mode[1] && kz == param->one_mode[2]))) {
                D[0] = D[1] = 0.0;
                nskip++;
            } else if (!v1rng) {
                if (nskip) {
                    zdo_pcg_advance(&v2rng[y], 0, (uint64_t) (2 * nskip));
                    nskip = 0;
                }
                cgauss2(Pk, kmag, &v2rng[y], D);
            } else {
                cgauss1(Pk, kmag, &v1rng[yres], D); /* only inside the k_cutoff region: :365-370 */
            }
            if (k2 == 0.0) k2 = 1.0;
            double ik2 = 1. / k2;

            if (gen_phi || input_phi_slab) { /* f_NL: src/zeldovich.cpp:377-400 */
                double H0 = 100., c = 299792.458;
                double growth = 1. / (1 + param->z_initial);
                double M = 2. * growth * c * c * zdo_infer_Tk(Pk, kmag) * k2 / (3. * param->Omega_M * H0 * H0);
                if (gen_phi) {
                    cset(AYZX(slab, 0, yres, z, x), D[0] / M, D[1] / M);
                    cset(AYZX(slabHer, 0, yresHer, zHer, xHer), D[0] / M, -D[1] / M);
                    continue;
                }
                if (kx == 0 && ky == 0 && kz == 0) {
                    D[0] = D[1] = 0.;
                } else {
                    const double *ph = AYZX_PHI((double *) input_phi_slab, 0, yres, z, x);
                    D[0] = ph[0] * M;
                    D[1] = ph[1] * M;
                }
            }

            if (D[0] != 0. || D[1] != 0.) {
                double e[4];
                zdo_get_eigenmode(eig, eig_ppd, kx, ky, kz, ppd, param->qPLT, e);
                double rescale = 1.;
                f              = 1.0;
                if (param->qPLT) {
                    f = (sqrt(1. + 24 * e[3] * param->f_cluster) - 1) * .25;
                    if (param->qPLTrescale) {
                        double plt_f = f;
                        rescale      = pow(a_NL / a0, target_f - plt_f);
                    }
                }
                /* F = rescale * I * e.vec[j] * fundamental * ik2 * D, evaluated left to right as
                 * std::complex arithmetic does: ((((rescale*I)*e)*fundamental)*ik2)*D   (:432-434) */
                double s;
                s    = rescale * e[0] * param->fundamental * ik2;
                F[0] = -s * D[1];
                F[1] = s * D[0];
                s    = rescale * e[1] * param->fundamental * ik2;
                G[0] = -s * D[1];
                G[1] = s * D[0];
                s    = rescale * e[2] * param->fundamental * ik2;
                H[0] = -s * D[1];
                H[1] = s * D[0];
            } else {
                F[0] = F[1] = G[0] = G[1] = H[0] = H[1] = 0.0;
                f                                      = 0.;
            }

            if (!just_density) {
                /* A = D + iF, B = G + iH   (:447-452) */
                cset(AYZX(slab, 0, yres, z, x), D[0] - F[1], D[1] + F[0]);
                cset(AYZX(slab, 1, yres, z, x), G[0] - H[1], G[1] + H[0]);
                if (param->qPLT) {
                    cset(AYZX(slab, 2, yres, z, x), 0. - F[1] * f, F[0] * f);
                    cset(AYZX(slab, 3, yres, z, x), G[0] * f - H[1] * f, G[1] * f + H[0] * f);
                }
                /* reflected entry: conj(D) + i conj(F) etc.   (:460-466) */
                cset(AYZX(slabHer, 0, yresHer, zHer, xHer), D[0] + F[1], -D[1] + F[0]);
                cset(AYZX(slabHer, 1, yresHer, zHer, xHer), G[0] + H[1], -G[1] + H[0]);
                if (param->qPLT) {
                    cset(AYZX(slabHer, 2, yresHer, zHer, xHer), 0. + F[1] * f, F[0] * f);
                    cset(AYZX(slabHer, 3, yresHer, zHer, xHer), G[0] * f + H[1] * f, -G[1] * f + H[0] * f);
                }
            } else {
                cset(AYZX(slab, 0, yres, z, x), D[0], D[1]);
                cset(AYZX(slabHer, 0, yresHer, zHer, xHer), D[0], -D[1]);
            }
        }
    }
    if (!v1rng) {
        zdo_pcg_advance(&v2rng[y], 0, (uint64_t) (2 * nskip));
        /* the reference's RNG bookkeeping self-check: :478 */
        assert(zdo_pcg_distance(&checkpoint, &v2rng[y]) == (uint64_t) (2 * ZDO_MAX_PPD * ZDO_MAX_PPD));
    }

    /* ky = 0: copy half of the reflected plane back, zero the origin   (:485-503) */
    if (yblock == 0 && yres == 0) {
        for (z = 0; z < ppdhalf; z++) {
            zHer = ppd - z;
            if (z == 0) zHer = 0;
            int xmax = (z == 0 ? ppdhalf : ppd);
            for (x = 0; x < xmax; x++) {
                xHer = ppd - x;
                if (x == 0) xHer = 0;
                for (int a = 0; a < g->narray; a++) {
                    double *src = AYZX(slabHer, a, yresHer, zHer, xHer);
                    cset(AYZX(slab, a, yres, zHer, xHer), src[0], src[1]);
                }
            }
        }
        for (int a = 0; a < g->narray; a++) cset(AYZX(slab, a, 0, 0, 0), 0.0, 0.0);
    }
}

/* BlockArray addressing: include/block_array.h:33-34, src/block_array.cpp:506-514 */
static inline double *blk_ptr(const geom *g, double *arr, int yblock, int zblock) {
    return arr + 2 * (((int64_t) zblock * g->numblock + yblock) * ((int64_t) g->block * g->block * g->ppd * g->narray));
}
/* StoreBlock: src/block_array.cpp:387-414 */
static void store_block(const geom *g, double *arr, int yblock, int zblock, const double *slab) {
    double *io = blk_ptr(g, arr, yblock, zblock);
    for (int a = 0; a < g->narray; a++)
        for (int zres = 0; zres < g->block; zres++)
            for (int yres = 0; yres < g->block; yres++) {
                int z = zres + g->block * zblock;
                memcpy(io, AYZX((double *) slab, a, yres, z, 0), sizeof(double) * 2 * (size_t) g->ppd);
                io += 2 * g->ppd;
            }
}
/* LoadBlock with the y shift: src/block_array.cpp:466-504 */
static void load_block(const geom *g, double *arr, int yblock, int zblock, double *slab) {
    double *io = blk_ptr(g, arr, yblock, zblock);
    for (int a = 0; a < g->narray; a++)
        for (int zres = 0; zres < g->block; zres++)
            for (int yres = 0; yres < g->block; yres++) {
                int y = yres + g->block * yblock, yshift;
                if (y >= g->ppdhalf)
                    yshift = y + 1;
                else
                    yshift = y;
                if (yshift == g->ppd) yshift = (int) g->ppdhalf;
                memcpy(AZYX(slab, a, zres, yshift, 0), io, sizeof(double) * 2 * (size_t) g->ppd);
                io += 2 * g->ppd;
            }
}

/* test helper: the same StoreBlock-all / LoadBlock-all sequence as oracle/ref_src/ref_blockarray.cpp drives through
 * the reference's BlockArray; tests/golden/blockarray_kat.json (made from the reference object code) pins it */
int zdo_blockarray_roundtrip(int ppd, int numblock, int narray, const double *slabs_in, double *arr_out,
                             double *slabs_out, double fill) {
    geom gg;
    gg.ppd = ppd;
    gg.ppdhalf = ppd / 2;
    gg.narray = narray;
    gg.numblock = numblock;
    gg.block = ppd / numblock;
    const geom *g = &gg;
    const int64_t slab_c = (int64_t) g->block * narray * ppd * ppd;
    for (int64_t i = 0; i < 2 * (int64_t) ppd * ppd * ppd * narray; i++) arr_out[i] = 0.0;
    for (int yblock = 0; yblock < numblock; yblock++)
        for (int zblock = 0; zblock < numblock; zblock++) store_block(g, arr_out, yblock, zblock, slabs_in + 2 * slab_c * yblock);
    for (int64_t i = 0; i < 2 * slab_c * numblock; i++) slabs_out[i] = fill;
    for (int zblock = 0; zblock < numblock; zblock++)
        for (int yblock = 0; yblock < numblock; yblock++) load_block(g, arr_out, yblock, zblock, slabs_out + 2 * slab_c * zblock);
    return 0;
}

typedef struct { unsigned short i, j, k, pad; double displ[3]; } zel_rec;               /* ZelParticle */
typedef struct { unsigned short i, j, k, pad; float displ[3]; float vel[3]; } rv_rec;   /* RVZelParticle */
typedef struct { unsigned short i, j, k, pad; double displ[3]; double vel[3]; } rvdouble_rec; /* RVdoubleZelParticle */

/* WriteParticlesSlab into memory: src/output.cpp:41-234 */
static void write_particles_slab(const geom *g, const zdo_params *param, int z, const double *s1,
                                 const double *s2, const double *s3, const double *s4, char *rec_out,
                                 float *dens_out, zdo_stats *st) {
    int just_density = param->qdensity == 2;
    double norm = 1.0, densitynorm = 1.0, vnorm;
    if (param->qPLT)
        vnorm = 1.0;
    else
        vnorm = (sqrt(1. + 24 * param->f_cluster) - 1) * .25;
    int recsize                 = zdo_record_size(param->icformat);
    double thisdensity_variance = 0.0;
    int64_t i                   = 0;
    for (int y = 0; y < g->ppd; y++) {
        for (int x = 0; x < g->ppd; x++) {
            int64_t yx  = (int64_t) x + g->ppd * y;
            double pos[3] = {0, 0, 0}, vel[3] = {0, 0, 0};
            double dens = s1[2 * yx] * densitynorm;
            if (!just_density) {
                pos[0] = s1[2 * yx + 1] * norm;
                pos[1] = s2[2 * yx] * norm;
                pos[2] = s2[2 * yx + 1] * norm;
                if (param->qPLT) {
                    vel[0] = s3[2 * yx + 1] * vnorm;
                    vel[1] = s4[2 * yx] * vnorm;
                    vel[2] = s4[2 * yx + 1] * vnorm;
                } else {
                    vel[0] = s1[2 * yx + 1] * vnorm;
                    vel[1] = s2[2 * yx] * vnorm;
                    vel[2] = s2[2 * yx + 1] * vnorm;
                }
                if (rec_out) {
                    /* record structs: include/output.h:19-42 (padding bytes, indeterminate in the
                     * reference, are written as zero) */
                    switch (param->icformat) {
                        case 2: {
                            rvdouble_rec *o = (rvdouble_rec *) rec_out + i;
                            o->i = (unsigned short) z; o->j = (unsigned short) y; o->k = (unsigned short) x; o->pad = 0;
                            o->displ[0] = pos[2]; o->displ[1] = pos[1]; o->displ[2] = pos[0];
                            o->vel[0] = vel[2]; o->vel[1] = vel[1]; o->vel[2] = vel[0];
                            break;
                        }
                        case 1: {
                            rv_rec *o = (rv_rec *) rec_out + i;
                            o->i = (unsigned short) z; o->j = (unsigned short) y; o->k = (unsigned short) x; o->pad = 0;
                            o->displ[0] = (float) pos[2]; o->displ[1] = (float) pos[1]; o->displ[2] = (float) pos[0];
                            o->vel[0] = (float) vel[2]; o->vel[1] = (float) vel[1]; o->vel[2] = (float) vel[0];
                            break;
                        }
                        case 0: {
                            zel_rec *o = (zel_rec *) rec_out + i;
                            o->i = (unsigned short) z; o->j = (unsigned short) y; o->k = (unsigned short) x; o->pad = 0;
                            o->displ[0] = pos[2]; o->displ[1] = pos[1]; o->displ[2] = pos[0];
                            break;
                        }
                        case 3: {
                            float *o = (float *) rec_out + 3 * i;
                            o[0] = (float) pos[2]; o[1] = (float) pos[1]; o[2] = (float) pos[0];
                            break;
                        }
                    }
                }
                for (int j = 0; j < 3; j++)
                    st->max_disp[j] = fabs(pos[j]) > fabs(st->max_disp[j]) ? pos[j] : st->max_disp[j];
            }
            if (param->qdensity && dens_out) dens_out[i] = (float) dens;
            thisdensity_variance += dens * dens;
            i++;
        }
    }
    st->density_variance += thisdensity_variance;
}

static int check_geom(const zdo_params *p, geom *g) {
    g->ppd      = p->ppd;
    g->ppdhalf  = p->ppd / 2;
    g->narray   = zdo_narray(p);
    g->numblock = p->numblock;
    /* src/block_array.cpp:38-40 */
    if (p->ppd % 2 != 0 || p->numblock % 2 != 0 || p->ppd % p->numblock != 0) return 1;
    g->block = (int) (p->ppd / p->numblock);
    return 0;
}

/* StoreBlockForward / LoadBlockForward: src/block_array.cpp:416-464 (block rows ordered [a][yres][zres]) */
static void store_block_forward(const geom *g, double *arr, int yblock, int zblock, const double *slab) {
    double *io = blk_ptr(g, arr, yblock, zblock);
    for (int a = 0; a < g->narray; a++)
        for (int yres = 0; yres < g->block; yres++)
            for (int zres = 0; zres < g->block; zres++) {
                int y = yres + g->block * yblock;
                memcpy(io, AZYX((double *) slab, a, zres, y, 0), sizeof(double) * 2 * (size_t) g->ppd);
                io += 2 * g->ppd;
            }
}
static void load_block_forward(const geom *g, double *arr, int yblock, int zblock, double *slab) {
    double *io = blk_ptr(g, arr, yblock, zblock);
    for (int a = 0; a < g->narray; a++)
        for (int yres = 0; yres < g->block; yres++)
            for (int zres = 0; zres < g->block; zres++) {
                int z = zres + g->block * zblock;
                memcpy(AYZX(slab, a, yres, z, 0), io, sizeof(double) * 2 * (size_t) g->ppd);
                io += 2 * g->ppd;
            }
}

/* ZeldovichZ: src/zeldovich.cpp:517-601 */
static int zeldovich_z(const geom *g, const zdo_params *param, const zdo_pk *Pk, zdo_pcg *v2rng, zdo_mt *v1rng, const double *eig,
                       int64_t eig_ppd, const fft_plan *pl, double *arr, int gen_phi, const geom *gphi,
                       double *phi_arr, zdo_stats *stats) {
    int64_t ppd = g->ppd;
    int n       = (int) ppd;
    int64_t len = (int64_t) g->block * ppd * ppd * g->narray;
    double *slab    = (double *) calloc((size_t) len, 2 * sizeof(double));
    double *slabHer = (double *) calloc((size_t) len, 2 * sizeof(double));
    double *input_phi_slab = NULL;
    if (phi_arr) input_phi_slab = (double *) calloc((size_t) gphi->block * ppd * ppd, 2 * sizeof(double));
    if (!slab || !slabHer || (phi_arr && !input_phi_slab)) return 2;
    for (int yblock = 0; yblock < g->numblock / 2; yblock++) {
        if (phi_arr) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int zblock = 0; zblock < gphi->numblock; zblock++) load_block_forward(gphi, phi_arr, yblock, zblock, input_phi_slab);
        }
        double t0 = now_sec();
#pragma omp parallel for schedule(dynamic, 1)
        for (int yres = 0; yres < g->block; yres++) {
            if (input_phi_slab) /* ForwardFFT_Yonly of the phi plane: :324-326 */
                forward_fft_first_index(pl, input_phi_slab + 2 * (int64_t) yres * ppd * ppd, n);
            load_plane_modes(g, param, Pk, v2rng, v1rng, eig, eig_ppd, yblock, yres, slab, slabHer, gen_phi, input_phi_slab);
            int yresHer = g->block - 1 - yres;
            for (int a = 0; a < g->narray; a++) { /* :508-511 */
                inverse_fft_first_index(pl, AYZX(slab, a, yres, 0, 0), n);
                inverse_fft_first_index(pl, AYZX(slabHer, a, yresHer, 0, 0), n);
            }
        }
        double t1 = now_sec();
#pragma omp parallel for schedule(dynamic, 1)
        for (int zblock = 0; zblock < g->numblock; zblock++) {
            store_block(g, arr, yblock, zblock, slab);
            store_block(g, arr, g->numblock - 1 - yblock, zblock, slabHer);
        }
        double t2 = now_sec();
        stats->t_stage1 += t1 - t0;
        stats->t_store += t2 - t1;
    }
    free(slabHer);
    free(slab);
    free(input_phi_slab);
    return 0;
}

/* ZeldovichXY_Phi: src/zeldovich.cpp:699-790 */
static int zeldovich_xy_phi(const geom *g, const zdo_params *param, const fft_plan *pl, double *arr) {
    int64_t ppd = g->ppd;
    int n       = (int) ppd;
    int64_t len = (int64_t) g->block * ppd * ppd * g->narray;
    double *slab = (double *) calloc((size_t) len, 2 * sizeof(double));
    if (!slab) return 2;
    double inv_ppd3 = 1. / ppd / ppd / ppd;
    for (int zblock = 0; zblock < g->numblock; zblock++) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int yblock = 0; yblock < g->numblock; yblock++) load_block(g, arr, yblock, zblock, slab);
        int ynyq = (int) (ppd / 2);
        for (int zres = 0; zres < g->block; zres++)
            for (int a = 0; a < g->narray; a++)
                for (int x = 0; x < ppd; x++) cset(AZYX(slab, a, zres, ynyq, x), 0.0, 0.0);
        for (int a = 0; a < g->narray; a++) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int zres = 0; zres < g->block; zres++) inverse_fft_2d(pl, AZYX(slab, a, zres, 0, 0), n);
        }
#pragma omp parallel for schedule(static)
        for (int zres = 0; zres < g->block; zres++)
            for (int y = 0; y < ppd; y++)
                for (int x = 0; x < ppd; x++) {
                    double *p  = AZYX(slab, 0, zres, y, x);
                    double phi = p[0];
                    cset(p, (phi + param->f_NL * phi * phi) * inv_ppd3, 0.0);
                }
        for (int a = 0; a < g->narray; a++) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int zres = 0; zres < g->block; zres++) forward_fft_2d(pl, AZYX(slab, a, zres, 0, 0), n);
        }
#pragma omp parallel for schedule(dynamic, 1)
        for (int yblock = 0; yblock < g->numblock; yblock++) store_block_forward(g, arr, yblock, zblock, slab);
    }
    free(slab);
    return 0;
}

int zdo_run(const zdo_params *param, const zdo_pk *Pk, const double *eig, int64_t eig_ppd, void *records,
            float *density, double *planes, zdo_stats *stats) {
    geom gg, *g = &gg;
    if (check_geom(param, g)) return 1;
#ifdef _OPENMP
    if (param->nthreads > 0) omp_set_num_threads(param->nthreads);
#endif
    memset(stats, 0, sizeof(*stats));
    int64_t ppd = g->ppd;
    int n       = (int) ppd;
    fft_plan *pl = fft_plan_create(n);
    zdo_pcg *v2rng = make_v2rng(param);
    zdo_mt *v1rng  = make_v1rng(param);

    /* ---- f_NL: phi field, local non-Gaussian transform (src/zeldovich.cpp:945-960) ---- */
    geom gphi = *g;
    gphi.narray = 1;
    double *phi_arr = NULL;
    if (param->f_NL != 0.) {
        int64_t tp = ppd * ppd * ppd;
        phi_arr    = (double *) calloc((size_t) tp, 2 * sizeof(double));
        if (!phi_arr) return 2;
        zdo_stats dummy;
        memset(&dummy, 0, sizeof(dummy));
        if (zeldovich_z(&gphi, param, Pk, v2rng, v1rng, eig, eig_ppd, pl, phi_arr, 1, NULL, NULL, &dummy)) return 2;
        if (zeldovich_xy_phi(&gphi, param, pl, phi_arr)) return 2;
    }

    int64_t total = ppd * ppd * ppd * g->narray;
    double *arr   = (double *) malloc((size_t) total * 2 * sizeof(double));
    if (arr) { /* BlockArray constructor zeroes in parallel, outside the stage timers (block_array.cpp:63-66) */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < 2 * total; i++) arr[i] = 0.0;
    }
    int64_t len  = (int64_t) g->block * ppd * ppd * g->narray;
    double *slab = (double *) calloc((size_t) len, 2 * sizeof(double));
    if (!arr || !slab) return 2;

    if (zeldovich_z(g, param, Pk, v2rng, v1rng, eig, eig_ppd, pl, arr, 0, &gphi, phi_arr, stats)) return 2;
    free(phi_arr);

    /* ---- ZeldovichXY: src/zeldovich.cpp:611-695 ---- */
    int recsize = zdo_record_size(param->icformat);
    for (int zblock = 0; zblock < g->numblock; zblock++) {
        double t0 = now_sec();
#pragma omp parallel for schedule(dynamic, 1)
        for (int yblock = 0; yblock < g->numblock; yblock++) load_block(g, arr, yblock, zblock, slab);
        double t1 = now_sec();
        int ynyq  = (int) (ppd / 2);
#pragma omp parallel for schedule(static)
        for (int zres = 0; zres < g->block; zres++)
            for (int a = 0; a < g->narray; a++)
                for (int x = 0; x < ppd; x++) cset(AZYX(slab, a, zres, ynyq, x), 0.0, 0.0);
        for (int a = 0; a < g->narray; a++) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int zres = 0; zres < g->block; zres++) inverse_fft_2d(pl, AZYX(slab, a, zres, 0, 0), n);
        }
        double t2 = now_sec();
        for (int zres = 0; zres < g->block; zres++) {
            int z = zres + g->block * zblock;
            if (planes)
                memcpy(planes + 2 * (int64_t) z * g->narray * ppd * ppd, AZYX(slab, 0, zres, 0, 0),
                       sizeof(double) * 2 * (size_t) (g->narray * ppd * ppd));
            if (param->qoneslab < 0 || z == param->qoneslab) {
                char *rec_out = records ? (char *) records + (size_t) z * (size_t) (ppd * ppd) * recsize : NULL;
                float *dens_out = density ? density + (size_t) z * (size_t) (ppd * ppd) : NULL;
                int last        = (int) (g->narray - 1);
                write_particles_slab(g, param, z, AZYX(slab, 0, zres, 0, 0),
                                     AZYX(slab, 1 < last ? 1 : last, zres, 0, 0),
                                     AZYX(slab, 2 < last ? 2 : last, zres, 0, 0),
                                     AZYX(slab, 3 < last ? 3 : last, zres, 0, 0), rec_out, dens_out, stats);
            }
        }
        double t3 = now_sec();
        stats->t_load += t1 - t0;
        stats->t_fft2d += t2 - t1;
        stats->t_write += t3 - t2;
    }
    free(slab);
    free(arr);
    free(v2rng);
    free(v1rng);
    fft_plan_destroy(pl);
    return 0;
}

/* Test helper: the mode cube as the XY stage sees it before any FFT (twin displacement undone,
 * Nyquist row zero); layout [a][y][z][x] */
int zdo_mode_cube(const zdo_params *param, const zdo_pk *Pk, const double *eig, int64_t eig_ppd,
                  double *cube) {
    geom gg, *g = &gg;
    if (check_geom(param, g)) return 1;
    int64_t ppd    = g->ppd;
    zdo_pcg *v2rng = make_v2rng(param);
    zdo_mt *v1rng  = make_v1rng(param);
    int64_t len    = (int64_t) g->block * ppd * ppd * g->narray;
    double *slab    = (double *) calloc((size_t) len, 2 * sizeof(double));
    double *slabHer = (double *) calloc((size_t) len, 2 * sizeof(double));
    memset(cube, 0, sizeof(double) * 2 * (size_t) (ppd * ppd * ppd * g->narray));
    for (int yblock = 0; yblock < g->numblock / 2; yblock++) {
        for (int yres = 0; yres < g->block; yres++)
            load_plane_modes(g, param, Pk, v2rng, v1rng, eig, eig_ppd, yblock, yres, slab, slabHer, 0, NULL);
        for (int yres = 0; yres < g->block; yres++) {
            for (int which = 0; which < 2; which++) {
                /* global y index as stored, then the LoadBlock shift (src/block_array.cpp:487-491) */
                int y = which == 0 ? yres + g->block * yblock : yres + g->block * (g->numblock - 1 - yblock);
                int yshift = y >= g->ppdhalf ? y + 1 : y;
                if (yshift == ppd) yshift = (int) g->ppdhalf;
                const double *src = which == 0 ? slab : slabHer;
                for (int a = 0; a < g->narray; a++)
                    memcpy(cube + 2 * ((int64_t) a * ppd + yshift) * ppd * ppd,
                           AYZX((double *) src, a, yres, 0, 0), sizeof(double) * 2 * (size_t) (ppd * ppd));
            }
        }
    }
    /* the Nyquist row is zeroed by ZeldovichXY (src/zeldovich.cpp:644-650) */
    for (int a = 0; a < g->narray; a++)
        memset(cube + 2 * ((int64_t) a * ppd + g->ppdhalf) * ppd * ppd, 0, sizeof(double) * 2 * (size_t) (ppd * ppd));
    free(slab);
    free(slabHer);
    free(v2rng);
    free(v1rng);
    return 0;
}

/* Counter-addressed single mode; pins the property the GPU generator relies on */
void zdo_mode_draw(const zdo_params *p, const zdo_pk *pk, int kx, int ky, int kz, uint64_t r[2],
                   double D[2]) {
    zdo_pcg g;
    unsigned long longseed = (unsigned long) (long) p->seed;
    zdo_pcg_seed(&g, (uint64_t) longseed);
    u128 c = 2 * (((u128) (uint64_t) ky * 65536 + (uint64_t) (kz & 65535)) * 65536 + (uint64_t) (kx & 65535));
    zdo_pcg_advance(&g, (uint64_t) (c >> 64), (uint64_t) c);
    zdo_pcg cp = g;
    r[0]       = zdo_pcg_next(&cp);
    r[1]       = zdo_pcg_next(&cp);
    double k2  = (kx * kx + ky * ky + kz * kz) * (p->fundamental * p->fundamental);
    cgauss2(pk, sqrt(k2), &g, D);
}

/* Direct summation of the displacement / velocity / density fields at a few lattice sites, without any FFT, blocking or
 * packing:   q_j(x) = sum_k F_j(k) e^{+2 pi i k.x / N},  F_j = I rescale e_j fundamental / k^2 D(k)   (src/zeldovich.cpp:404-452:
 * pos[0] = Im(A) = the inverse transform of F etc.; src/output.cpp:93-141 for the velocity: PLT f F_j, else vnorm q_j),
 * density = sum_k D(k) e^{...}.  The modes are drawn exactly as LoadPlane draws them — plane ky's stream walked in (z, x)
 * order, zeroed modes skipped in bulk (src/zeldovich.cpp:335-363) — for the half space ky in [0, N/2); a mode and its
 * Hermitian twin contribute 2 Re[F e^{i theta}].  In the ky = 0 plane the reference keeps the draws of z in [0, N/2) (x < N/2
 * for z = 0) and overwrites the others with their conjugates, origin zeroed (:485-503): only those "winner" modes are summed.
 * Requires every Nyquist-plane mode to be zeroed by the rule (no CornerModes), version 2 streams, f_NL = 0.
 * The phases are exact: theta = 2 pi m / N with m = k.x mod N from a table of N entries.  Kahan sums, one partial sum per
 * ky plane, added in ky order (result independent of the thread count).
 *   sites: nsites x (z, y, x);   out: nsites x 7 = qx, qy, qz, vx, vy, vz, density  (this code's x, y, z order).
 * A size-independent oracle check of the GPU path at the full BASELINE sizes (tests/golden/direct_sum_*.json). */
int zdo_direct_sum(const zdo_params *param, const zdo_pk *Pk, const double *eig, int64_t eig_ppd, int nsites, const int *sites,
                   double *out) {
    const int64_t ppd = param->ppd, ppdhalf = ppd / 2;
    if (param->CornerModes || param->version == 1 || param->f_NL != 0. || nsites < 1 || nsites > 64) return 1;
    const double fundamental2 = param->fundamental * param->fundamental;
    const double ik_cutoff    = 1.0 / param->k_cutoff;
    const double target_f     = (sqrt(1. + 24 * param->f_cluster) - 1) / 4.;
    const double a_NL = param->qPLTrescale ? 1. / (1 + param->PLT_target_z) : 1.0;
    const double a0   = param->qPLTrescale ? 1. / (1 + param->z_initial) : 1.0;
    const double k2_cutoff = param->nyquist * param->nyquist / (param->k_cutoff * param->k_cutoff);
    const int kmax = (double) ppdhalf * ik_cutoff + .5;
    const double vnorm = param->qPLT ? 1.0 : (sqrt(1. + 24 * param->f_cluster) - 1) * .25;
    double *cs = (double *) malloc(sizeof(double) * 2 * (size_t) ppd);
    for (int64_t m = 0; m < ppd; m++) {
        long double a = 2.0L * 3.141592653589793238462643383279502884L * (long double) m / (long double) ppd;
        cs[2 * m]     = (double) cosl(a);
        cs[2 * m + 1] = (double) sinl(a);
    }
    zdo_pcg *v2rng   = make_v2rng(param);
    const int NQ     = 7;
    double *partial  = (double *) calloc((size_t) ppdhalf * nsites * NQ, sizeof(double));
    int bad = 0;
    int nthreads = param->nthreads > 0 ? param->nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t y = 0; y < ppdhalf; y++) {
        const int ky = (int) y;
        double sum[64 * 7], comp[64 * 7];
        for (int i = 0; i < nsites * NQ; i++) sum[i] = comp[i] = 0.0;
        zdo_pcg rng   = v2rng[y];
        int64_t nskip = 0;
        int64_t mrow[64], mstep[64];
        for (int z = 0; z < ppd; z++) {
            if (z == ppdhalf + 1) nskip += (ZDO_MAX_PPD - ppd) * ZDO_MAX_PPD;
            const int kz = z > ppdhalf ? z - (int) ppd : z;
            for (int s = 0; s < nsites; s++) {
                /* phase index of kx = 0 in this row, and its step per unit of kx */
                int64_t m = ((int64_t) ky * sites[3 * s + 1] + (int64_t) kz * sites[3 * s]) % ppd;
                mrow[s]  = m < 0 ? m + ppd : m;
                mstep[s] = sites[3 * s + 2] % ppd;
            }
            for (int x = 0; x < ppd; x++) {
                if (x == ppdhalf + 1) nskip += ZDO_MAX_PPD - ppd;
                const int kx = x > ppdhalf ? x - (int) ppd : x;
                double k2    = (kx * kx + ky * ky + kz * kz) * fundamental2;
                if ((abs(kx) == kmax || abs(kz) == kmax || abs(ky) == kmax) || k2 >= k2_cutoff
                    || (param->qonemode && !(kx == param->one_mode[0] && ky == param->one_mode[1] && kz == param->one_mode[2]))) {
                    nskip++;
                    continue;
                }
                if (abs(kx) == ppdhalf || abs(kz) == ppdhalf) {  /* a live Nyquist-plane mode: not Hermitian, unsupported */
#pragma omp atomic write
                    bad = 1;
                }
                if (nskip) {
                    zdo_pcg_advance(&rng, 0, (uint64_t) (2 * nskip));
                    nskip = 0;
                }
                double D[2];
                cgauss2(Pk, sqrt(k2), &rng, D);
                if (ky == 0) { /* winners of the ky = 0 plane (:485-503); the origin is zeroed */
                    const int winner = (z < ppdhalf) && (z > 0 || (x > 0 && x < ppdhalf));
                    if (!winner) continue;
                }
                if (D[0] == 0. && D[1] == 0.) continue;
                if (k2 == 0.0) k2 = 1.0;
                const double ik2 = 1. / k2;
                double e[4];
                zdo_get_eigenmode(eig, eig_ppd, kx, ky, kz, ppd, param->qPLT, e);
                double rescale = 1., f = 1.0;
                if (param->qPLT) {
                    f = (sqrt(1. + 24 * e[3] * param->f_cluster) - 1) * .25;
                    if (param->qPLTrescale) rescale = pow(a_NL / a0, target_f - f);
                }
                double sj[3];
                for (int j = 0; j < 3; j++) sj[j] = rescale * e[j] * param->fundamental * ik2;
                const double fv = param->qPLT ? f : vnorm;
                for (int s = 0; s < nsites; s++) {
                    int64_t m = (mrow[s] + (int64_t) kx * mstep[s]) % ppd;
                    if (m < 0) m += ppd;
                    const double c = cs[2 * m], sn = cs[2 * m + 1];
                    /* 2 Re[D e^{i t}] and 2 Re[i D e^{i t}] */
                    const double dre = 2.0 * (D[0] * c - D[1] * sn);
                    const double fim = -2.0 * (D[0] * sn + D[1] * c);
                    double term[7];
                    for (int j = 0; j < 3; j++) {
                        term[j]     = sj[j] * fim;
                        term[3 + j] = fv * (sj[j] * fim);
                    }
                    term[6] = dre;
                    for (int q = 0; q < NQ; q++) { /* Kahan */
                        double *S = &sum[s * NQ + q], *Cc = &comp[s * NQ + q];
                        const double yv = term[q] - *Cc;
                        const double tv = *S + yv;
                        *Cc = (tv - *S) - yv;
                        *S  = tv;
                    }
                }
            }
        }
        for (int i = 0; i < nsites * NQ; i++) partial[(size_t) y * nsites * NQ + i] = sum[i];
    }
    for (int i = 0; i < nsites * NQ; i++) {
        double S = 0.0, Cc = 0.0;
        for (int64_t y = 0; y < ppdhalf; y++) {
            const double yv = partial[(size_t) y * nsites * NQ + i] - Cc;
            const double tv = S + yv;
            Cc = (tv - S) - yv;
            S  = tv;
        }
        out[i] = S;
    }
    free(partial);
    free(v2rng);
    free(cs);
    return bad ? 2 : 0;
}
