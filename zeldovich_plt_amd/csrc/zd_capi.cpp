// zd_capi.cpp — C ABI (include/zeldovich_hip.h): plan management, the single-GPU driver
// zd_generate (= ZeldovichZ + ZeldovichXY, src/zeldovich.cpp:517-695) and the device test hooks.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/zeldovich_hip.h"
#include "zd_device.h"
#include "zd_launch.h"
#include "zd_plan.h"
#ifdef ZD_TUNING
#include "zd_tuning.h"
#endif
#ifdef ZD_TESTING
#include "zd_testing.h"
#endif

using zdfft::cplx;
using zdpcg::u128;

#define HIPCHECK(call)                                                                               \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            fprintf(stderr, "zeldovich_hip: %s failed at %s:%d: %s\n", #call, __FILE__, __LINE__,    \
                    hipGetErrorString(e__));                                                         \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

// ---- dispatch diagnostics (zd_launch.h DispatchSite) ----
namespace zd {
static std::atomic<DispatchSite *> g_dispatch_head{nullptr};
DispatchSite::DispatchSite(const char *f, int l) : func(f), line(l) {
    DispatchSite *h = g_dispatch_head.load(std::memory_order_relaxed);
    do {
        next = h;
    } while (!g_dispatch_head.compare_exchange_weak(h, this, std::memory_order_release, std::memory_order_relaxed));
}
}  // namespace zd

int64_t zd_dispatch_report(char *buf, int64_t cap) {
    int64_t need = 0, w = 0;  // bytes the whole report takes / bytes written: whole lines only, nothing after the first that does not fit
    for (zd::DispatchSite *s = zd::g_dispatch_head.load(std::memory_order_acquire); s; s = s->next) {
        char line[1024];
        int n = snprintf(line, sizeof line, "%lld\t%d\t%s\n", s->count.load(std::memory_order_relaxed), s->line, s->func);
        if (n <= 0) continue;
        if (n >= (int) sizeof line) {  // (a launcher name longer than the line: snprintf returns what it WOULD have written)
            n = (int) sizeof line - 1;
            line[n - 1] = '\n';
        }
        if (buf && w == need && w + n < cap) {
            memcpy(buf + w, line, (size_t) n);
            w += n;
        }
        need += n;
    }
    if (buf && cap > 0) buf[w] = 0;  // (w < cap: a C string of at most cap - 1 bytes, every byte of it written)
    return need + 1;
}

// Large device buffers the kernels only partly write (stores, exchange rings, phi fields).  In the -DZD_TESTING library
// zd_test_poison(1) makes every such allocation start out as NaN bytes, so that a kernel reading something no kernel wrote
// shows up in the parity tests instead of depending on what a fresh hipMalloc happens to contain.
#ifdef ZD_TESTING
static std::atomic<int> g_poison{0};
void zd_test_poison(int on) { g_poison.store(on); }
#endif
hipError_t zd_store_alloc(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
#ifdef ZD_TESTING
    if (e == hipSuccess && g_poison.load()) e = hipMemset(*p, 0xFF, bytes);
#endif
    return e;
}

namespace {


bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// Tuning / ablation knobs (environment variables ZD_ABLATE, ZD_PRUNE, ZD_NT, ZD_LAYOUT, ...) exist only in the
// -DZD_TUNING build (make tuning -> build/libzeldovich_hip_tuning.so, used by scripts/ through ZD_LIB_PATH).
// The product library never reads the environment: the names below do not even reach its binary.
#ifdef ZD_TUNING
const char *tune_env(const char *name) { return getenv(name); }
#else
#define tune_env(name) ((const char *) nullptr)
#endif

std::vector<cplx> make_twiddles(int n) {
    std::vector<cplx> tw(n);
    for (int k = 0; k < n; k++) {
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double) k / (long double) n;
        tw[k].x = (double) cosl(a);
        tw[k].y = (double) sinl(a);
    }
    return tw;
}

// Bluestein tables of a length-n transform on the engine of size M = 2^m >= 2n - 1 (zd_kernels_any.hip): c_m = exp(i pi m^2/n)
// with m^2 reduced mod 2n in integers, and the forward DFT_M of b_m = conj(c_m), |m| < n, wrapped (long double radix-2)
int any_make_tab(int n, zd::AnyTab *tab, cplx **d_bufs) {
    const int M = zd::any_engine_size(n);
    const long double PI = 3.14159265358979323846264338327950288L;
    std::vector<cplx> chirp(n), tw = make_twiddles(M);
    std::vector<long double> br(M, 0.0L), bi(M, 0.0L);
    for (long long m = 0; m < n; m++) {
        const long double a = PI * (long double) ((m * m) % (2LL * n)) / (long double) n;
        const long double c = cosl(a), sn = sinl(a);
        chirp[m] = cplx{(double) c, (double) sn};
        br[m] = c;
        bi[m] = -sn;
        if (m) {
            br[M - m] = c;
            bi[M - m] = -sn;
        }
    }
    // forward DFT (sign -1), iterative radix-2
    for (int i = 1, j = 0; i < M; i++) {
        int bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(br[i], br[j]);
            std::swap(bi[i], bi[j]);
        }
    }
    for (int len = 2; len <= M; len <<= 1) {
        const long double ang = -2.0L * PI / (long double) len;
        for (int i = 0; i < M; i += len)
            for (int k = 0; k < len / 2; k++) {
                const long double wr = cosl(ang * k), wi = sinl(ang * k);
                const long double ur = br[i + k], ui = bi[i + k];
                const long double vr = br[i + k + len / 2] * wr - bi[i + k + len / 2] * wi;
                const long double vi = br[i + k + len / 2] * wi + bi[i + k + len / 2] * wr;
                br[i + k] = ur + vr;
                bi[i + k] = ui + vi;
                br[i + k + len / 2] = ur - vr;
                bi[i + k + len / 2] = ui - vi;
            }
    }
    std::vector<cplx> fb(M);
    for (int i = 0; i < M; i++) fb[i] = cplx{(double) br[i], (double) bi[i]};
    const std::vector<cplx> *src[3] = {&tw, &chirp, &fb};
    for (int i = 0; i < 3; i++) {
        if (hipMalloc((void **) &d_bufs[i], sizeof(cplx) * src[i]->size()) != hipSuccess) return 1;
        if (hipMemcpy(d_bufs[i], src[i]->data(), sizeof(cplx) * src[i]->size(), hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    tab->n     = n;
    tab->M     = M;
    tab->invM  = 1.0 / (double) M;
    tab->twM   = d_bufs[0];
    tab->chirp = d_bufs[1];
    tab->fb    = d_bufs[2];
    return 0;
}

// LDS image of k_genf; offsets must match GenfTab (zd_kernels.hip): directions {cos, sin}(2 pi j/512) at 0,
// ln bins {c_j, -ln c_j} at 1024, 2^(j/64) at 1392, spline segment records at 1456, uint16 segment LUT after them
std::vector<double> build_genf_table(const zd_pk *pk, int nseg, double *lut_x0, double *lut_inv_dx) {
    constexpr int SC = 0, LG = 1024, EX = 1392, SEG = 1456, GLUT = 1024;
    std::vector<double> T(SEG + 6 * (size_t) nseg + GLUT / 4, 0.0);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int j = 0; j < 512; j++) {
        T[SC + 2 * j]     = (double) cosl(2.0L * pi * j / 512.0L);
        T[SC + 2 * j + 1] = (double) sinl(2.0L * pi * j / 512.0L);
    }
    for (int j = 0; j < 182; j++) {  // f in [(j+181)/256, (j+182)/256); bins 74, 75 surround f = 1
        double c = 1.0, l = 0.0;
        if (j != 74 && j != 75) {
            c = (double) (1.0L / (((long double) (j + 181) + 0.5L) / 256.0L));
            l = (double) (-logl((long double) c));
        }
        T[LG + 2 * j]     = c;
        T[LG + 2 * j + 1] = l;
    }
    for (int j = 0; j < 64; j++) T[EX + j] = (double) powl(2.0L, (long double) j / 64.0L);
    *lut_x0 = 0.0;
    *lut_inv_dx = 0.0;
    if (nseg > 0) {
        for (int i = 0; i < nseg; i++) {  // SplineFunction::val, include/spline_function.h:153-162
            const double h = pk->x[i + 1] - pk->x[i];
            double *r = &T[SEG + 6 * (size_t) i];
            r[0] = pk->x[i];
            r[1] = 1.0 / h;
            r[2] = pk->y[i];
            r[3] = pk->y[i + 1];
            r[4] = pk->y2[i] * (h * h) / 6.0;
            r[5] = pk->y2[i + 1] * (h * h) / 6.0;
        }
        // segment start table: cell c covers ln k in [x0 + c*dx, x0 + (c+1)*dx); never past the true start
        unsigned short *lut = reinterpret_cast<unsigned short *>(&T[SEG + 6 * (size_t) nseg]);
        const double x0 = pk->x[0], x1 = pk->x[nseg];
        const double dx = (x1 - x0) / GLUT;
        int klo = 0;
        for (int c = 0; c < GLUT; c++) {
            const double cell_start = x0 + c * dx * (1.0 - 1e-12) - 1e-9;
            while (klo < nseg - 1 && pk->x[klo + 1] <= cell_start) klo++;
            lut[c] = (unsigned short) klo;
        }
        *lut_x0     = x0;
        *lut_inv_dx = 1.0 / dx;
    }
    return T;
}

int record_size(int icformat) {  // include/output.h:19-42
    switch (icformat) {
        case ZD_FMT_ZEL: return 32;
        case ZD_FMT_RVZEL: return 32;
        case ZD_FMT_RVDOUBLEZEL: return 56;
        case ZD_FMT_ZELSIMPLE: return 12;
    }
    return 0;
}

}  // namespace

namespace {

void tick(zd_plan *pl, int kind, hipStream_t st, bool begin) {
    if (!pl->p.profile) return;
    if (begin) {
        EventPair ep;
        hipEventCreate(&ep.a);
        hipEventCreate(&ep.b);
        ep.kind = kind;
        hipEventRecord(ep.a, st);
        pl->events.push_back(ep);
    } else {
        hipEventRecord(pl->events.back().b, st);
    }
}

// span events (begin and end recorded at different places, other ticks in between)
int span_begin(zd_plan *pl, int kind, hipStream_t st) {
    if (!pl->p.profile) return -1;
    EventPair ep;
    hipEventCreate(&ep.a);
    hipEventCreate(&ep.b);
    ep.kind = kind;
    hipEventRecord(ep.a, st);
    pl->events.push_back(ep);
    return (int) pl->events.size() - 1;
}
void span_end(zd_plan *pl, int idx, hipStream_t st) {
    if (idx >= 0) hipEventRecord(pl->events[idx].b, st);
}

void collect_events(zd_plan *pl) {
    for (auto &ep : pl->events) {
        float ms = 0;
        hipEventSynchronize(ep.b);
        hipEventElapsedTime(&ms, ep.a, ep.b);
        pl->kernel_ms[ep.kind] += ms;
        pl->launches[ep.kind]++;
        hipEventDestroy(ep.a);
        hipEventDestroy(ep.b);
    }
    pl->events.clear();
}

// Rows of the block store are N*16 B apart: at PPD >= 2048 that is a multiple of 32 KB and the strided accesses of
// the y pass pile onto a few HBM channels (PPD=4096: k_yfft 1.20 s -> 0.92 s with the pad, PPD=2048: 100 -> 93 ms).
// 384 B of padding per row de-aliases them (128 B and 640 B do as well; 256 B does not at PPD=2048) for 0.6-5 % more memory.
int store_row_pad(int64_t N) { return N >= 512 ? 24 : 0; }

bool p_oneslab_off(const zd_plan *pl) { return pl->p.qoneslab < 0; }  // ZD_qoneslab runs finish a single pass

int64_t y_bytes_per_row(const zd_plan *pl) { return (int64_t) pl->jobs.n * pl->L * pl->N * 16; }

// advance by (2*65536*drows - 1) draws, drows may be negative (period 2^128)
zdpcg::Affine row_jump(long long drows) {
    const __int128 d = (__int128) 2 * 65536 * (__int128) drows - 1;
    return zdpcg::jump_map((u128) d);
}
zdpcg::Affine row_jump_full(long long drows) {  // 2*65536*drows draws
    const __int128 d = (__int128) 2 * 65536 * (__int128) drows;
    return zdpcg::jump_map((u128) d);
}

}  // namespace

extern "C" {

// ZD_qdensity = 1 on the composite grids (PPD = 2^a 3^b 5^c 7^d; round 4): the ZA field store carries two more half-space sums — the
// density D of the two residues of a pass — and the y / x stages add one array, delta_r0 + i delta_r1 (zd_kernels_np2.hip).  The
// power-of-two grids keep the reference's arrays for ZD_qdensity (their kernels exist).  PLT with ZD_qdensity = 1 on a composite grid
// is composed of a PLT run and density-only passes of this store (plt_dens_split); f_NL stays on the convolution path there.
static bool dens_fields(const zd_params *p) {
    // (8640 = 64 * 135 included: the six-field store runs there too — tests/test_gpu_baseline_regime.py,
    // test_density_one_mode_at_every_composite_size[8640]; the PPD > 8192 gate's message says so)
    // (ZD_qdensity = 2, density only — round 5: the same six-field store, the displacement arrays are simply not built and no records
    // are written; the four potentials ride along unused, which still beats the convolution path by ~4x)
    return (p->qdensity == 1 || p->qdensity == 2) && !p->qPLT && p->f_NL == 0. && p->qoneslab < 0 && !is_pow2(p->ppd) && zd::np2_supported_ppd((int) p->ppd)
           && (p->store_mode == ZD_STORE_AUTO || p->store_mode == ZD_STORE_FIELDS);
}
// Packed stores (zd_device.h PACK_*): without ZD_qdensity the density field is not transformed.
static int pack_mode(const zd_params *p, int R) {
    if (p->store_mode == ZD_STORE_REFERENCE) return zd::PACK_NONE;
    if ((p->qdensity != 0 && !dens_fields(p)) || p->f_NL != 0.) return zd::PACK_NONE;
    if (p->qoneslab >= 0) return zd::PACK_NONE;  // density_variance is then the sum over that one slab (output.cpp:197)
    {   // The packed stores treat every field as the transform of a REAL field (Hermitian modes) and take
        // density_variance from sum |D|^2.  That needs every mode with a component on the Nyquist plane |k_i| = N/2 to
        // be zero: the |k_i| == kmax rule does it when kmax == N/2 (k_cutoff = 1), the spherical cut when k_cutoff >= 1
        // and CornerModes is off.  Otherwise (e.g. CornerModes with k_cutoff = 2) the reference keeps independent,
        // non-Hermitian draws there and takes Re/Im of the mixed field (zeldovich.cpp:350-356): reference arrays.
        const int half = (int) (p->ppd / 2), kmax = (int) ((double) half * (1.0 / p->k_cutoff) + .5);
        const bool nyquist_dead = kmax == half || (!p->corner_modes && p->k_cutoff >= 1.0);
        if (!nyquist_dead) return zd::PACK_NONE;
    }
    // PPD = 8192: three lines of a row no longer fit a workgroup of the x pass; only the field store has an x kernel that
    // takes them in sequence (k_xfft_seq)
    // PLT: the three packed arrays by default; its field store (six half-space sums) only on request — measured slower
    // (PPD=2048 PLT+rescale 0.553 -> 0.609 s: every array of the y stage needs two potentials, each fetched twice, and the
    // 148-VGPR PLT generator leaves no room for the z FFT beside it anyway)
    // (PPD = 8192: the x pass of the packed arrays does not exist — three lines of a row are 1536 threads — so PLT runs on
    // its field store there, whose ring goes through k_xfft_two)
    // PPD = 2^a 3^b: the composite-transform kernels exist for the field stores only
    if (p->qPLT) return (p->ppd > 4096 || !is_pow2(p->ppd) || p->store_mode == ZD_STORE_FIELDS) ? zd::PACK_PLTFIELD : zd::PACK_PLT3;
    if (R < 2) return zd::PACK_NONE;  // the ZA packings carry two z-residues per pass
    if (p->store_mode == ZD_STORE_PACKED) return p->ppd > 4096 ? zd::PACK_NONE : zd::PACK_ZAPAIR;
    return zd::PACK_ZAFIELD;
}
static int store_arrays(const zd_params *p, int R) {
    if (pack_mode(p, R) != zd::PACK_NONE) return 3;
    return p->qdensity == 2 ? 1 : (p->qPLT ? 4 : 2);
}

// the zero rule of zeldovich.cpp:350-353 as the kernels prune by it (zd_device.h column_is_zero)
static void prune_rule(const zd_params *p, zd::StoreLayout &S) {
    const int half = (int) (p->ppd / 2);
    S.N         = (int) p->ppd;
    S.half      = half;
    S.prune     = 7;
    S.kmax      = (int) ((double) half * (1.0 / p->k_cutoff) + .5);
    S.fund2     = p->fundamental * p->fundamental;
    S.k2_cutoff = p->corner_modes ? 0.0 : p->nyquist * p->nyquist / (p->k_cutoff * p->k_cutoff);  // CornerModes: only |k_i| == kmax prunes
}

// Block table of the field store (zd_device.h FieldLayout) for `nranks` ranks: block b stands for the row slots 8b .. 8b+7
// (rows ky = c + nranks*slot) of every rank c and is sized by the longest of them (ky = nranks*8b).  Returns the elements
// per (plane, field) image.
static int64_t field_rows(const zd_params *p, int nranks, std::vector<zd::FieldRow> *rows) {
    zd::StoreLayout S;
    memset(&S, 0, sizeof(S));
    prune_rule(p, S);
    const int N = S.N, half = S.half, Hq = half / nranks, CW = zd::FIELD_CW, NT = std::max(1, N / CW), RB = zd::FIELD_RB;
    const int nblk = (Hq + RB - 1) / RB;
    int64_t total = 0;
    if (rows) rows->resize(nblk);
    for (int i = 0; i < nblk; i++) {
        const int ky = i * RB * nranks;
        // live columns are kx in (-c, c): tiles [0, tl) on the low side, [th, NT) on the high side
        int tl = 0, th = NT;
        if (N >= 2 * CW) {
            while (tl < NT) {  // tile tl holds a live column?
                bool live = false;
                for (int x = tl * CW; x < (tl + 1) * CW && !live; x++) live = !zd::column_is_zero(S, x > half ? x - N : x, ky);
                if (!live) break;
                tl++;
            }
            while (th > tl) {
                bool live = false;
                for (int x = (th - 1) * CW; x < th * CW && !live; x++) live = !zd::column_is_zero(S, x > half ? x - N : x, ky);
                if (!live) break;
                th--;
            }
            // a live tile strictly between dead ones cannot happen for the symmetric interval rule, but the table must be
            // safe for any rule: if one exists, keep the whole row
            for (int tt = tl; tt < th; tt++)
                for (int x = tt * CW; x < (tt + 1) * CW; x++)
                    if (!zd::column_is_zero(S, x > half ? x - N : x, ky)) { tl = NT; th = NT; }
        } else {
            tl = NT;
        }
        zd::FieldRow r;
        r.base  = (int) total;
        r.split = (unsigned short) (tl >= th ? N : tl * CW);
        r.gap   = (unsigned short) (tl >= th ? 0 : (th - tl) * CW);
        if (rows) (*rows)[i] = r;
        // 384 B of padding per row as in the other stores: the first ~N/11 rows are complete (64 KiB apart at PPD = 4096)
        total += (int64_t) (N - r.gap + store_row_pad(N)) * RB;
    }
    return total;
}

// bytes of the block store one rank holds per pass (send side; nranks > 1 doubles it with the receive buffer)
static int64_t store_bytes(const zd_params *p, int R, int nranks) {
    const int64_t N = p->ppd;
    const int pm = pack_mode(p, R);
    if (zd::pack_is_fields(pm)) return (N / R) * (pm == zd::PACK_PLTFIELD ? 6 : (dens_fields(p) ? 6 : 4)) * field_rows(p, nranks, nullptr) * 16;
    return N * (N + store_row_pad(N)) * (N / R) / nranks * 16 * store_arrays(p, R);
}
// ring between the y and x stages of the field store: planes of the three PACK_ZAPAIR arrays
static int field_ring_planes(int64_t N, int64_t Zq) {
    const int64_t plane_b = 3 * N * (N + store_row_pad(N)) * 16;
    return (int) std::max<int64_t>(1, std::min<int64_t>(Zq, ((int64_t) 6 << 30) / plane_b));
}

// stream factors the composite (2^a 3^b 5^c 7^d) kernels take: any EVEN divisor of PPD (two residues r, r + R/2 share a ZA pass) — or
// 1 — whose z lines have a composite transform.  R = 36 gives PPD = 6912 z lines of 192 = 64 * 3 and 18 passes where the powers of two
// offer 64 (z lines of 108) and 32 passes: the store of a pass must fit, and between 128 GB and 260 GB there was nothing.
static bool np2_stream_factor_ok(int64_t N, int R) {
    return R >= 1 && (R == 1 || R % 2 == 0) && N % R == 0 && zd::np2_supported_zlen((int) (N / R));
}

// ZD_qdensity = 2 (density only): the displacement arrays are never built (src/zeldovich.cpp:303,440), so the eigenmodes of ZD_qPLT
// never enter — the run IS the ZA density-only run, on every grid and path
static zd_params canonical(const zd_params *p) {
    zd_params c = *p;
    if (c.qdensity == 2) c.qPLT = c.qPLTrescale = 0;
    return c;
}

// PLT with ZD_qdensity = 1 on the composite grids, one rank (round 5): the PLT field store has no density field and its paired
// generator no registers for a seventh pair of sums, so the density planes come from a second plan — density only, ZA six-field
// store (dens_only) — at stream factor 2R: its pass j holds the residues j and j + R, i.e. exactly the planes z = j (mod R) of this
// plan's pass j, in the same delivery order.  It runs at the head of the Z stage on the same store (which the PLT pass then
// overwrites) and leaves N/R planes of float32 behind.  Before: the ~6x slower convolution path.
static bool plt_dens_split(const zd_params *p, int nranks) {
    return p->qPLT && p->qdensity == 1 && p->f_NL == 0. && p->qoneslab < 0 && nranks == 1 && !is_pow2(p->ppd)
           && zd::np2_supported_ppd((int) p->ppd) && (p->store_mode == ZD_STORE_AUTO || p->store_mode == ZD_STORE_FIELDS)
           && !tune_env("ZD_NO_DENS_SPLIT");
}
// (the two halves of such a run)
static zd_params split_plt_half(const zd_params *p) {
    zd_params a = *p;
    a.qdensity  = 0;
    return a;
}
static zd_params split_dens_half(const zd_params *p, int R) {
    zd_params b     = *p;
    b.qPLT          = 0;
    b.qPLTrescale   = 0;
    b.qdensity      = 2;
    b.stream_factor = 2 * R;
    return b;
}

static int choose_stream_factor_one(const zd_params *p, int nranks, int64_t budget_bytes);
int zd_choose_stream_factor(const zd_params *p_in, int nranks, int64_t budget_bytes) {
    const zd_params pc = canonical(p_in);
    if (plt_dens_split(&pc, nranks)) {
        // the PLT half's own choice, with room for what the density half adds: N/R planes of float32, its two rings (3 + 1 arrays of
        // <= 6 GB / 3) and its tables; both R and 2R must have z lines the composite kernels transform
        const zd_params pa = split_plt_half(&pc);
        const int64_t N = pc.ppd;
        for (int R = 1; N / R >= 24; R = R == 1 ? 2 : R + 2) {
            if (!np2_stream_factor_ok(N, R) || !np2_stream_factor_ok(N, 2 * R) || N / R > 2048) continue;
            if (!zd::pack_is_fields(pack_mode(&pa, R)) || (N / 2) % zd::FIELD_RB) break;
            const int64_t ring = (int64_t) field_ring_planes(N, N / R) * N * (N + store_row_pad(N)) * 16;
            const int64_t need = store_bytes(&pa, R, 1) + 3 * ring                                             // the PLT half
                                 + (N / R) * N * N * 4 + 4 * ring + ((int64_t) 1 << 30);                       // + the density half
            if (need <= budget_bytes) return R;
        }
    }
    return choose_stream_factor_one(&pc, nranks, budget_bytes);
}
static int choose_stream_factor_one(const zd_params *p, int nranks, int64_t budget_bytes) {
    const int64_t N = p->ppd;
    // ZA without density: two residues share a pass, so R = 2 is preferred over R = 1 whenever the z FFT is long enough
    int R0 = 1;
    if (!p->qPLT && pack_mode(p, 2) != zd::PACK_NONE && N / 2 >= 32 && (N / 2) % nranks == 0) R0 = 2;
    const bool np2 = !is_pow2(N);
    if (N > 8192 && pack_mode(p, 2) != zd::PACK_ZAFIELD) return -1;  // 16384, 8640: only the ZA field store has kernels (plan_create_ex)
    if (!np2 && pack_mode(p, 2) == zd::PACK_NONE && (p->qdensity == 2 ? 1 : (p->qPLT ? 4 : 2)) * (N / 16) > 1024) return -1;  // 8192 on four arrays: no x pass
    // any other even PPD (or a 2^a 3^b one whose options the composite kernels lack): reference arrays on one rank, see
    // plan_create_ex; R any divisor of N
    auto any_factor = [&]() -> int {
        if (N % 2 || N < 8 || N > 8192 || nranks != 1) return -1;
        const int64_t narray = p->qdensity == 2 ? 1 : (p->qPLT ? 4 : 2);
        for (int R = 1; N / R >= 3; R++)  // any divisor of N (the z-residue fold is a plain decimation: R need not be 2^k here)
            if (N % R == 0 && (N / R) * narray * N * (N + store_row_pad(N)) * 16 <= budget_bytes) return R;
        return -1;
    };
    if (np2 && (!zd::pack_is_fields(pack_mode(p, 2)) || !zd::np2_supported_ppd((int) N) || (N / 2) % (nranks * zd::FIELD_RB))) return any_factor();
    for (int R = R0; np2 ? N / R >= 12 : (N / R >= 32 && N % R == 0); R = np2 ? (R == 1 ? 2 : R + 2) : R * 2) {
        if (np2) {  // any even divisor whose z lines have a composite transform (np2_stream_factor_ok)
            if (!np2_stream_factor_ok(N, R) || (N / R) % nranks) continue;
        } else if ((N / R) % nranks)
            break;
        if (N / R > 4096) continue;  // z-FFT kernels exist up to length 4096
        if (N > 4096 && N / R > 2048) continue;  // the field store's z FFT stops at 2048 and PPD > 4096 has no other store worth using
        int64_t store = store_bytes(p, R, nranks);
        if (nranks > 1) store += std::min<int64_t>(store, (int64_t) 9 << 30);  // + the two-slot exchange ring (zd_multi.cpp), not a second store
        if (zd::pack_is_fields(pack_mode(p, R)))  // + the y -> x ring (3 arrays; 4 with the density array)
            store += (int64_t) field_ring_planes(N, N / R / nranks) * (dens_fields(p) ? 4 : 3) * N * (N + store_row_pad(N)) * 16;
        if (store <= budget_bytes) return R;
    }
    return np2 ? any_factor() : -1;
}

// a job that one rank runs as convolutions on the power-of-two engine (zd_kernels_any.hip: PPD neither 2^a nor a composite size
// whose options the composite kernels have).  Its R is any divisor of PPD; several GPUs share it as pass groups only.
static bool convolution_job(const zd_params *p) {
    const int64_t N = p->ppd;
    if (is_pow2(N) || plt_dens_split(p, 1)) return false;
    return p->f_NL != 0. || !zd::pack_is_fields(pack_mode(p, 2)) || !zd::np2_supported_ppd((int) N) || (N / 2) % zd::FIELD_RB;
}

// planes a store plane delivers for (p, R) on `nranks` ranks: 2 with the ZA packings (plan_create_ex's choice of store)
static int plan_plane_step(const zd_params *p, int R, int nranks) {
    const int64_t N = p->ppd;
    if (!is_pow2(N)) {  // composite kernels (field stores) or the convolution path (reference arrays)
        const bool comp = zd::np2_supported_ppd((int) N) && p->f_NL == 0. && np2_stream_factor_ok(N, R)
                          && zd::pack_is_fields(pack_mode(p, R)) && (N / 2) % (nranks * zd::FIELD_RB) == 0;
        if (!comp) return 1;
    }
    int pm = pack_mode(p, R);
    if (zd::pack_is_fields(pm) && (((N / 2) / nranks) % zd::FIELD_RB || N / R > 2048))
        pm = N > 4096 ? zd::PACK_NONE : (pm == zd::PACK_PLTFIELD ? zd::PACK_PLT3 : zd::PACK_ZAPAIR);
    return (pm == zd::PACK_ZAPAIR || pm == zd::PACK_ZAFIELD) ? 2 : 1;
}

int zd_choose_pass_groups(const zd_params *p_in, int ngpu, int64_t budget_bytes, int32_t *groups, int32_t *stream_factor) {
    const zd_params pc = canonical(p_in), *p = &pc;
    if (ngpu < 1) ngpu = 1;
    int g = p->pass_groups;
    if (g < 0 || (g > 0 && ngpu % g)) {
        fprintf(stderr, "zeldovich_hip: ZD_PassGroups = %d does not divide ZD_NumGPU = %d\n", g, ngpu);
        return 1;
    }
    if (p->f_NL != 0.) g = g > 0 ? g : 1;  // the phi round is one collective job
    if (g > 1 && p->f_NL != 0.) {
        fprintf(stderr, "zeldovich_hip: ZD_f_NL != 0 runs as one group of ranks (ZD_PassGroups = 1)\n");
        return 1;
    }
    if (g == 0) {  // automatic: one GPU per group — no exchange — while a single rank's job has, or can be given, that many passes
        g = 1;
        zd_params q = *p;
        const int R1 = q.stream_factor > 0 ? q.stream_factor : zd_choose_stream_factor(&q, 1, budget_bytes);
        if (R1 > 0 && ngpu > 1) {
            // A free stream factor may be doubled (smaller stores always fit): every rank then generates the modes once per
            // pass of its own instead of sharing one generation, and nothing travels.  PPD = 4096 ZA on 8 GPUs: R = 8 -> 16,
            // one pass per GPU, 0.36-0.38 s predicted against 0.30-0.33 s for the all-to-all — within the
            // uncertainty of the link rate, and independent of it.  More than one doubling is not worth the generations.
            const int Rmax = q.stream_factor > 0 ? R1 : 2 * R1;
            if (convolution_job(p) && p->f_NL == 0.) {  // any divisor of PPD is a stream factor there; up to four times the passes it needs
                for (int R2 = R1; R2 <= (q.stream_factor > 0 ? R1 : std::max(4 * R1, 2 * ngpu)) && p->ppd / R2 >= 3; R2++)
                    if (p->ppd % R2 == 0 && R2 % ngpu == 0) {
                        g = ngpu;
                        break;
                    }
            } else
            for (int R2 = R1; R2 <= Rmax; R2 = is_pow2(p->ppd) ? R2 * 2 : R2 + 2) {
                const bool len_ok = is_pow2(p->ppd) ? (p->ppd % R2 == 0 && p->ppd / R2 >= 32) : np2_stream_factor_ok(p->ppd, R2);
                if (!len_ok) {
                    if (is_pow2(p->ppd)) break;
                    continue;
                }
                const int npass = R2 / plan_plane_step(&q, R2, 1);
                if (npass >= ngpu && npass % ngpu == 0) {
                    g = ngpu;
                    break;
                }
            }
        }
    }
    const int gsz = ngpu / g;
    int R = p->stream_factor > 0 ? p->stream_factor : zd_choose_stream_factor(p, gsz, budget_bytes);
    if (R < 0) return 1;
    // Ranks that exchange pipeline their passes (zd_plan_run_passes: Z stage of pass p+1 beside the exchange of pass p, two
    // send stores): give every group at least four passes when the stream factor is free.  A larger factor always fits —
    // the stores shrink — and costs (passes / ranks per group) generations per rank, hidden behind the exchange.
    if (gsz > 1 && p->stream_factor <= 0 && is_pow2(p->ppd))
        while ((R / plan_plane_step(p, R, gsz)) < 4 * g && p->ppd % (2 * R) == 0 && p->ppd / (2 * R) >= 256 && (p->ppd / (2 * R)) % gsz == 0) R *= 2;
    // the passes must deal out evenly over the groups: a larger stream factor (smaller stores) always fits
    while ((R / plan_plane_step(p, R, gsz)) % g) {
        if (p->stream_factor > 0) {
            fprintf(stderr, "zeldovich_hip: ZD_StreamFactor = %d gives %d passes, not a multiple of the %d pass groups\n", R,
                    R / plan_plane_step(p, R, gsz), g);
            return 1;
        }
        if (is_pow2(p->ppd)) {
            R *= 2;
            if (p->ppd % R || p->ppd / R < 32 || (p->ppd / R) % gsz) return 1;
        } else if (gsz == 1 && convolution_job(p)) {  // convolution kernels: the next divisor of PPD
            do R++;
            while (p->ppd / R >= 3 && p->ppd % R);
            if (p->ppd / R < 3) return 1;
        } else {  // composite PPD: the next even factor the composite kernels take
            do R += 2;
            while (p->ppd / R >= 12 && !(np2_stream_factor_ok(p->ppd, R) && (p->ppd / R) % gsz == 0));
            if (p->ppd / R < 12) return 1;
        }
    }
    *groups        = g;
    *stream_factor = R;
    return 0;
}

// The choice with a measured link rate (zd_comm_probe).  Where zd_choose_pass_groups takes one GPU per pass group — nothing
// travels, every GPU generates all the modes once per pass of its own — the single group with the all-to-all is the alternative:
// the ranks share the generations (rows ky = rank mod G), the block store is transposed between the z and the y / x passes
// (src/block_array.cpp:387-414,466-504) in plane groups over every link of a GPU at once, passes pipelined over two send stores.
// Both are priced from unit times of ONE MI355X per particle of the grid (measured, DESIGN.md 4 / 5 — PPD = 4096: ZA a full
// generation 0.141 s, the z FFT beside it +0.185 s / 4 passes... per particle below; PLT from the PPD = 4096 PLT run; composite
// grids: y + x at 1.85x) and the exchange's bytes per link at the measured rate:
//   pass groups:  T = c_gen N^3 P / G + (c_zf + c_xy) N^3 / G                         P passes in all, P / G per GPU
//   all-to-all:   T = t_z / P' + max(t_ex, t_xy + t_z (P' - 1) / P') + 0.05 t_xy      t_z = (c_gen P' + c_zf) N^3 / G, t_xy = c_xy N^3 / G,
//                                                                                     t_ex = store bytes per rank and pass x P' / G / rate
// (first Z stage exposed, the others and the XY stages beside the exchange, the last plane group's XY behind it; P' = 1: t_z + max).
int zd_choose_pass_groups_measured(const zd_params *p_in, int ngpu, int64_t budget_bytes, double link_GBps, int32_t *groups,
                                   int32_t *stream_factor, double *est_seconds) {
    const zd_params pc = canonical(p_in), *p = &pc;
    if (est_seconds) est_seconds[0] = est_seconds[1] = 0.0;
    if (zd_choose_pass_groups(p, ngpu, budget_bytes, groups, stream_factor)) return 1;
    if (!(link_GBps > 0.0) || ngpu < 2 || p->pass_groups != 0 || *groups != ngpu) return 0;  // nothing to decide
    zd_params q   = *p;
    q.pass_groups = 1;
    int32_t g1 = 0, R1 = 0;
    if (zd_choose_pass_groups(&q, ngpu, budget_bytes, &g1, &R1) || g1 != 1) return 0;  // no single group for this job: keep
    const double n3 = (double) p->ppd * (double) p->ppd * (double) p->ppd;
    const bool comp = !is_pow2(p->ppd);
    // (ZD_k_cutoff = c keeps 1 / c^3 of the modes and 1 / c^2 of the (kx, ky) columns: generation and z FFT shrink with them,
    // the y / x stages and the records do not)
    const double kc = p->k_cutoff > 1.0 ? p->k_cutoff : 1.0;
    const double c_gen = (p->qPLT ? 3.9e-12 : 2.05e-12) / (kc * kc * kc), c_zf = (p->qPLT ? 3.5e-12 : 2.7e-12) / (kc * kc);
    const double c_xy  = (p->qPLT ? 3.7e-11 : 2.0e-11) * (comp ? 1.85 : 1.0);
    const int P0 = *stream_factor / plan_plane_step(p, *stream_factor, 1), P1 = R1 / plan_plane_step(&q, R1, ngpu);
    const double t_pg = c_gen * n3 * P0 / ngpu + (c_zf + c_xy) * n3 / ngpu;
    const double t_z = (c_gen * P1 + c_zf) * n3 / ngpu, t_xy = c_xy * n3 / ngpu;
    const double t_ex = (double) store_bytes(&q, R1, ngpu) * P1 / ngpu / (link_GBps * 1e9);
    const double t_a2a = P1 >= 2 ? t_z / P1 + std::max(t_ex, t_xy + t_z * (P1 - 1) / P1) + 0.05 * t_xy : t_z + std::max(t_ex, t_xy) + 0.05 * t_xy;
    if (est_seconds) {
        est_seconds[0] = t_pg;
        est_seconds[1] = t_a2a;
    }
    if (t_a2a < t_pg) {
        *groups        = 1;
        *stream_factor = R1;
    }
    return 0;
}

// phi_mode 1: first f_NL pass (one array holding phi = D/M); phik != NULL: second pass (D = phik * M)
static int plan_create_ex(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks,
                          int phi_mode, const cplx *phik, zd_plan **out);

// f_NL: phi field -> local transform phi + f_NL phi^2 -> Fourier space again (zeldovich.cpp:945-960, 699-790): the first
// ZeldovichZ / ZeldovichXY_Phi round of the reference, run once; *d_phik = PhiK[ky][kz][x] for the half-space rows (owned
// by the caller).  One rank: the forward z transform needs every plane resident.
static int make_phik(const zd_params *p, const zd_pk *pk, cplx **d_phik) {
    const int64_t N  = p->ppd;
    zd_params pp     = *p;
    pp.stream_factor = 1;
    pp.qPLT          = 0;  // the phi pass never reaches the displacement algebra
    zd_plan *ph      = nullptr;
    if (plan_create_ex(&pp, pk, nullptr, 0, 0, 1, 1, nullptr, &ph)) return 1;
    void *d_phi = nullptr;
    int frc     = 1;
    *d_phik     = nullptr;
    do {
        if (zd_store_alloc(&d_phi, (size_t) zd_plan_exchange_bytes(ph)) != hipSuccess
            || zd_store_alloc((void **) d_phik, (size_t) (N / 2) * N * N * 16) != hipSuccess) {
            fprintf(stderr, "zeldovich_hip: f_NL needs %.1f GB of HBM for the phi field at PPD %lld\n",
                    (zd_plan_exchange_bytes(ph) + (N / 2) * N * N * 16) / 1e9, (long long) N);
            break;
        }
        fprintf(stderr, "Generating phi field\n");
        if (zd_plan_stage_z(ph, 0, d_phi, 0) || zd_plan_stage_y(ph, d_phi, 0)) break;
        if (ph->any) {  // convolution-transform PPDs: x inverse, phi + f_NL phi^2, then the forward 3-D transform of that REAL
                        // field as the conjugate of its inverse transform (x, y, z in place; conjugated while PhiK is laid out)
            const long long plane = (long long) N * ph->AL.pitch;
            if (zd::launch_any_lines(ph->tabN, d_phi, ph->AL.pitch, (long long) N * N, 0)) break;
            if (zd::launch_any_phi_nl(ph->AL, p->f_NL, d_phi, 0)) break;
            if (zd::launch_any_lines(ph->tabN, d_phi, ph->AL.pitch, (long long) N * N, 0)) break;
            if (zd::launch_any_cols(ph->tabN, d_phi, plane, ph->AL.pitch, (int) N, (int) N, -1, 0)) break;
            if (zd::launch_any_cols(ph->tabN, d_phi, ph->AL.pitch, plane, (int) N, (int) (N / 2), -1, 0)) break;  // along z, rows ky < N/2
            if (zd::launch_any_phik(ph->AL, d_phi, *d_phik, 0)) break;
            if (hipDeviceSynchronize() != hipSuccess) break;
            frc = 0;
            break;
        }
        int lN = 0;
        while ((1 << lN) < (int) N) lN++;
        if (zd::launch_fnl_stage(0, ph->S, p->f_NL, ph->d_twN, d_phi, nullptr, (int) N, lN, 0)) break;
        if (zd::launch_fnl_stage(1, ph->S, p->f_NL, ph->d_twN, d_phi, nullptr, (int) N, lN, 0)) break;
        if (zd::launch_fnl_stage(2, ph->S, p->f_NL, ph->d_twN, d_phi, *d_phik, (int) N, lN, 0)) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        frc = 0;
    } while (0);
    hipFree(d_phi);
    zd_plan_destroy(ph);
    if (frc) {
        hipFree(*d_phik);
        *d_phik = nullptr;
    }
    return frc;
}

static zd::StoreLayout layout_for_chunks(const zd_plan *pl, int chunk_planes);

// ---- f_NL on several ranks (zd_multi.cpp): the plans of the phi round and of the main pass, and the stages of the phi round
// on a plane group / on the returned store ----
int zd_plan_create_phi(const zd_params *p, const zd_pk *pk, int rank, int nranks, zd_plan **out) {
    zd_params pp     = *p;
    pp.stream_factor = 1;  // the forward z transform needs every plane of a row: one pass
    pp.qPLT          = 0;
    return plan_create_ex(&pp, pk, nullptr, 0, rank, nranks, 1, nullptr, out);
}
int zd_plan_create_phik(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks, const void *d_phik,
                        zd_plan **out) {
    return plan_create_ex(p, pk, eig, eig_ppd, rank, nranks, 0, (const cplx *) d_phik, out);
}
// planes [0, nplanes) of a ring slot with `chunk_planes` planes per chunk: inverse y, x (inverse, phi + f_NL phi^2, forward),
// forward y — ZeldovichXY_Phi (zeldovich.cpp:699-790)
int zd_plan_phi_xy_group(zd_plan *pl, void *d_slot, int chunk_planes, int nplanes, double f_NL, void *hip_stream) {
    hipStream_t st = (hipStream_t) hip_stream;
    const zd::StoreLayout S = layout_for_chunks(pl, chunk_planes);
    if (zd::launch_yfft(S, nplanes, pl->d_twN, d_slot, st)) return 1;
    if (zd::launch_fnl_stage(0, S, f_NL, pl->d_twN, d_slot, nullptr, nplanes, 0, st)) return 1;
    return zd::launch_fnl_stage(1, S, f_NL, pl->d_twN, d_slot, nullptr, nplanes, 0, st);
}
// forward z over this rank's rows of the returned store -> PhiK[row slot][kz][x]
int zd_plan_phi_zfwd(zd_plan *pl, void *d_store, void *d_phik, void *hip_stream) {
    int lZq = 0;
    while ((1 << lZq) < pl->Zq) lZq++;
    return zd::launch_fnl_stage(2, pl->S, 0.0, pl->d_twN, d_store, d_phik, pl->N, lZq, (hipStream_t) hip_stream);
}

int zd_plan_create(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks,
                   zd_plan **out) {
    if (p->f_NL == 0.) return plan_create_ex(p, pk, eig, eig_ppd, rank, nranks, 0, nullptr, out);
    // ZD_f_NL: the phi round runs here, once; the plan owns PhiK and its Z stages read D = PhiK * M
    if (nranks != 1) {
        fprintf(stderr, "zeldovich_hip: ZD_f_NL != 0 runs on one rank (the forward z transform of the phi field needs every plane)\n");
        return 1;
    }
    cplx *d_phik = nullptr;
    if (make_phik(p, pk, &d_phik)) return 1;
    if (plan_create_ex(p, pk, eig, eig_ppd, 0, 1, 0, d_phik, out)) {
        hipFree(d_phik);
        return 1;
    }
    (*out)->d_phik_owned = d_phik;
    return 0;
}

static int plan_create_one(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks, int phi_mode,
                           const cplx *phik, zd_plan **out);

static int plan_create_ex(const zd_params *p_in, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks,
                          int phi_mode, const cplx *phik, zd_plan **out) {
    const zd_params pc = canonical(p_in);
    const bool split_R_ok = pc.stream_factor <= 0 || (np2_stream_factor_ok(pc.ppd, pc.stream_factor) && np2_stream_factor_ok(pc.ppd, 2 * pc.stream_factor));
    if (phi_mode == 0 && phik == nullptr && plt_dens_split(&pc, nranks) && split_R_ok && eig != nullptr && eig_ppd > 0) {
        const zd_params pa = split_plt_half(&pc);
        zd_plan *A = nullptr, *B = nullptr;
        bool ok = plan_create_one(&pa, pk, eig, eig_ppd, rank, nranks, 0, nullptr, &A) == 0;
        ok = ok && A->pack == zd::PACK_PLTFIELD && !A->any && np2_stream_factor_ok(pc.ppd, 2 * A->R);
        if (ok) {
            const zd_params pb = split_dens_half(&pc, A->R);
            ok = plan_create_one(&pb, pk, nullptr, 0, rank, nranks, 0, nullptr, &B) == 0;
            ok = ok && B->dens_only && B->npass == A->npass && B->pstep == 2 && 2 * B->Zq == A->Zq;
        }
        if (ok) ok = zd_store_alloc((void **) &A->d_dens_pass, (size_t) A->Zq * pc.ppd * pc.ppd * sizeof(float)) == hipSuccess;
        if (ok) {
            A->dens_sub     = B;
            A->p            = pc;  // (what the callers read back: ZD_qdensity = 1)
            A->store_bytes_ = std::max(A->store_bytes_, B->store_bytes_);
            *out            = A;
            return 0;
        }
        zd_plan_destroy(B);
        zd_plan_destroy(A);  // the run falls back to the reference arrays (convolution path)
    }
    return plan_create_one(&pc, pk, eig, eig_ppd, rank, nranks, phi_mode, phik, out);
}

static int plan_create_one(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks,
                           int phi_mode, const cplx *phik, zd_plan **out) {
    const int64_t N = p->ppd;
    // Three transform families: powers of two (zd_kernels.hip), 2^a 3^b on the field stores (zd_kernels_np2.hip), and ANY
    // other even PPD — or a 2^a 3^b one with options the composite kernels lack — as convolutions on the power-of-two engine
    // (zd_kernels_any.hip: reference arrays, one rank, no f_NL)
    const bool pow2 = is_pow2(N);
    bool any_path = false;
    int np2_R = 2;
    if (!pow2) {
        int Rg = p->stream_factor > 0 ? p->stream_factor : 2;
        if (p->stream_factor <= 0)  // no factor given: the first one whose z lines the composite kernels have
            for (int r = 2; N / r >= 12; r += 2)
                if (np2_stream_factor_ok(N, r)) {
                    Rg = r;
                    break;
                }
        np2_R = Rg;
        const bool comp_ok = zd::np2_supported_ppd((int) N) && phi_mode == 0 && phik == nullptr && np2_stream_factor_ok(N, Rg)
                             && zd::pack_is_fields(pack_mode(p, Rg))
                             && (N / 2) % (nranks * zd::FIELD_RB) == 0;
        any_path = !comp_ok;
    }
    const bool np2 = !pow2 && !any_path;  // PPD = 2^a 3^b: the composite-transform kernels, field stores only
    if (pow2 && (N < 32 || N > 16384)) {
        fprintf(stderr, "zeldovich_hip: PPD = %lld unsupported (powers of two: 32 ... 16384)\n", (long long) N);
        return 1;
    }
    if (p->qPLT && (eig == NULL || eig_ppd <= 0)) {
        fprintf(stderr, "zeldovich_hip: ZD_qPLT set but no eigenmode table given\n");
        return 1;
    }
    int R = p->stream_factor > 0 ? p->stream_factor : 1;
    if (np2 && p->stream_factor <= 0) R = np2_R;
    if (any_path) {
        if (N % 2 || N < 8 || N > 8192 || nranks != 1 || R < 1 || N % R || N / R < 3) {
            fprintf(stderr, "zeldovich_hip: PPD = %lld (neither 2^a nor a supported 2^a 3^b configuration) runs as convolutions on the "
                            "power-of-two engine: even PPD in [8, 8192], one rank, ZD_StreamFactor any divisor of PPD (got %d)\n",
                    (long long) N, R);
            return 1;
        }
    } else if (np2 ? !np2_stream_factor_ok(N, R) : (!is_pow2(R) || N % R || N / R < 32 || N / R > 4096)) {
        fprintf(stderr, "zeldovich_hip: stream factor %d invalid for PPD %lld\n", R, (long long) N);
        return 1;
    }
    if (nranks < 1 || !is_pow2(nranks) || rank < 0 || rank >= nranks || (N / 2) % nranks || (N / R) % nranks) {
        fprintf(stderr, "zeldovich_hip: cannot split PPD %lld (R=%d) over %d ranks (a power of two dividing PPD/2 and PPD/R is required)\n",
                (long long) N, R, nranks);
        return 1;
    }
    // legacy mt19937 streams, one per yres (power_spectrum.cpp:18-25); the second f_NL pass takes D from PhiK and draws nothing
    const bool v1 = p->version == 1 && phik == nullptr;
    if (p->version != 0 && p->version != 1 && p->version != 2) {
        fprintf(stderr, "zeldovich_hip: ZD_Version = %d (1 or 2 expected)\n", p->version);
        return 1;
    }
    if (v1 && (p->numblock <= 0 || N % p->numblock || (N / p->numblock) % nranks)) {
        fprintf(stderr, "zeldovich_hip: ZD_Version = 1 needs ZD_NumBlock dividing PPD and PPD/NumBlock streams divisible by the number of "
                        "ranks (%d)\n", nranks);
        return 1;
    }
    zd_plan *pl = new zd_plan;
    pl->p       = *p;
    pl->rank    = rank;
    pl->nranks  = nranks;
    pl->N       = (int) N;
    pl->half    = (int) (N / 2);
    // (density only: one array — zeldovich.cpp:871-876 — except on the composite kernels, where it rides on the six-field ZA store)
    pl->narray  = (phi_mode == 1 || (p->qdensity == 2 && (any_path || !dens_fields(p)))) ? 1 : (p->qPLT ? 4 : 2);
    if (phi_mode == 0 && phik == nullptr && pl->narray >= 2 && !any_path) pl->pack = pack_mode(p, R);
    pl->any = any_path;
    if (zd::pack_is_fields(pl->pack) && ((pl->half / nranks) % zd::FIELD_RB || N / R > 2048))  // row blocks of 8, z FFT <= 2048
        pl->pack = N > 4096 ? zd::PACK_NONE : (pl->pack == zd::PACK_PLTFIELD ? zd::PACK_PLT3 : zd::PACK_ZAPAIR);
    if (N > 8192 && pl->pack != zd::PACK_ZAFIELD) {  // a 16384-point line fills a workgroup: only the field store's kernels exist (also 8640 = 64 * 135: ZA)
        fprintf(stderr, "zeldovich_hip: PPD = %lld runs on the ZA field store only (ZD_StreamFactor >= 16; no ZD_qPLT / ZD_f_NL / ZD_qdensity = 2; "
                        "ZD_qdensity = 1 on the composite grid 8640 only)\n",
                (long long) N);
        delete pl;
        return 1;
    }
    if (pl->pack == zd::PACK_NONE && pow2 && (int64_t) pl->narray * (N / 16) > 1024) {
        // (launch_xfft_t would say so only after the Z and y stages of the first pass had run)
        fprintf(stderr, "zeldovich_hip: PPD = %lld on the reference's %d arrays (PLT with ZD_qdensity, ZD_f_NL or ZD_StoreMode = reference) has no "
                        "x pass: the lines of a row are %lld threads\n", (long long) N, pl->narray, (long long) pl->narray * (N / 16));
        delete pl;
        return 1;
    }
    if (pl->pack != zd::PACK_NONE) pl->narray = 3;
    pl->dens    = pl->pack == zd::PACK_ZAFIELD && dens_fields(p) && np2;
    pl->dens_only = pl->dens && p->qdensity == 2;  // density planes only: no displacement arrays, no records (src/output.cpp:94,207)
    if (pl->pack == zd::PACK_ZAFIELD && dens_fields(p) && !np2) {  // (cannot happen: dens_fields is composite-only)
        delete pl;
        return 1;
    }
    pl->pstep   = (pl->pack == zd::PACK_ZAPAIR || pl->pack == zd::PACK_ZAFIELD) ? 2 : 1;
    pl->npass   = R / pl->pstep;
    pl->R       = R;
    pl->L       = (int) (N / R);
    pl->Hq      = pl->half / nranks;
    pl->Zq      = pl->L / nranks;

    // ---- generator constants (zeldovich.cpp:299-320, 350) ----
    zd::GenConst &g = pl->g;
    memset(&g, 0, sizeof(g));
    g.N            = pl->N;
    g.half         = pl->half;
    g.kmax         = (int) ((double) pl->half * (1.0 / p->k_cutoff) + .5);
    g.corner_modes = p->corner_modes;
    g.ablate       = tune_env("ZD_ABLATE") ? atoi(tune_env("ZD_ABLATE")) : 0;
    g.qonemode     = p->qonemode;
    for (int i = 0; i < 3; i++) g.one_mode[i] = p->one_mode[i];
    g.fundamental  = p->fundamental;
    g.fundamental2 = p->fundamental * p->fundamental;
    g.k2_cutoff    = p->nyquist * p->nyquist / (p->k_cutoff * p->k_cutoff);
    {  // the comparison `k2 >= k2_cutoff` of zeldovich.cpp:353 is monotonic in the integer |k|^2
        long long c = (long long) floor(g.k2_cutoff / g.fundamental2);
        while (c > 0 && (double) c * g.fundamental2 >= g.k2_cutoff) c--;
        while ((double) c * g.fundamental2 < g.k2_cutoff) c++;
        g.k2i_cut = (int) std::min<long long>(c, 0x7fffffffLL);
    }
    g.pk_n         = pk->n;
    g.fixed_power  = pk->fixed_power;
    g.is_powerlaw  = pk->is_powerlaw;
    g.pk_norm      = pk->normalization;
    g.pk_smooth2   = pk->Pk_smooth2;
    g.powerlaw_index = pk->powerlaw_index;
    g.qPLT         = p->qPLT;
    g.qPLTrescale  = p->qPLTrescale;
    g.f_cluster    = p->f_cluster;
    g.target_f     = (sqrt(1. + 24 * p->f_cluster) - 1) / 4.;
    {
        double a_NL = 1.0, a0 = 1.0;
        if (p->qPLTrescale) {
            a_NL = 1. / (1 + p->PLT_target_z);
            a0   = 1. / (1 + p->z_initial);
        }
        g.ln_growth_ratio = log(a_NL / a0);
    }
    g.eig_ppd = eig_ppd;
    g.gen_phi = phi_mode == 1;
    g.phik    = phik;
    if (phi_mode == 1) g.qPLT = 0;  // the phi pass stops before the displacement algebra (zeldovich.cpp:385-391)

#define PLCHECK(call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            fprintf(stderr, "zeldovich_hip: %s failed at %s:%d: %s\n", #call, __FILE__, __LINE__, \
                    hipGetErrorString(e__));                                                      \
            zd_plan_destroy(pl);                                                                  \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

    if (!pk->is_powerlaw) {
        PLCHECK(hipMalloc((void **) &pl->d_pk, sizeof(double) * 3 * (size_t) pk->n));
        PLCHECK(hipMemcpy(pl->d_pk, pk->x, sizeof(double) * pk->n, hipMemcpyHostToDevice));
        PLCHECK(hipMemcpy(pl->d_pk + pk->n, pk->y, sizeof(double) * pk->n, hipMemcpyHostToDevice));
        PLCHECK(hipMemcpy(pl->d_pk + 2 * pk->n, pk->y2, sizeof(double) * pk->n, hipMemcpyHostToDevice));
        g.pk_x  = pl->d_pk;
        g.pk_y  = pl->d_pk + pk->n;
        g.pk_y2 = pl->d_pk + 2 * pk->n;
        // segment start table: cell c covers ln k in [x0 + c*dx, x0 + (c+1)*dx)
        std::vector<int> lut(zd::PK_LUT);
        const double x0 = pk->x[0], x1 = pk->x[pk->n - 1];
        const double dx = (x1 - x0) / zd::PK_LUT;
        int klo = 0;
        for (int c2 = 0; c2 < zd::PK_LUT; c2++) {
            const double cell_start = x0 + c2 * dx * (1.0 - 1e-12) - 1e-9;  // never past the true start
            while (klo < pk->n - 2 && pk->x[klo + 1] <= cell_start) klo++;
            lut[c2] = klo;
        }
        PLCHECK(hipMalloc((void **) &pl->d_lut, sizeof(int) * lut.size()));
        PLCHECK(hipMemcpy(pl->d_lut, lut.data(), sizeof(int) * lut.size(), hipMemcpyHostToDevice));
        g.pk_lut     = pl->d_lut;
        g.lut_x0     = x0;
        g.lut_inv_dx = 1.0 / dx;
    }
    if (p->qPLT) {
        const size_t nb = sizeof(double) * (size_t) eig_ppd * eig_ppd * (eig_ppd / 2 + 1) * 4;
        PLCHECK(hipMalloc((void **) &pl->d_eig, nb));
        PLCHECK(hipMemcpy(pl->d_eig, eig, nb, hipMemcpyHostToDevice));
        g.eig = pl->d_eig;
    }
    // per-plane stream heads == v2rng[] of the reference (power_spectrum.cpp:26-37)
    {
        std::vector<u128> rows(pl->half);
        const zdpcg::Affine plane = zdpcg::jump_map((u128) 2 * ZD_MAX_PPD * ZD_MAX_PPD);
        rows[0] = zdpcg::seed_state((uint64_t) p->seed);
        for (int i = 1; i < pl->half; i++) rows[i] = zdpcg::apply(plane, rows[i - 1]);
        PLCHECK(hipMalloc((void **) &pl->d_rowstate, sizeof(u128) * rows.size()));
        PLCHECK(hipMemcpy(pl->d_rowstate, rows.data(), sizeof(u128) * rows.size(), hipMemcpyHostToDevice));
        g.row_state = pl->d_rowstate;
        // (built once; ZD_NumGPU creates its plans from several host threads at a time)
        static const zdpcg::BitTable bt = [] {
            zdpcg::BitTable b;
            zdpcg::make_bit_table(b);
            return b;
        }();
        if (zdk_upload_bit_table(&bt) != 0 || zdk_upload_bit_table_fz(&bt) != 0) {
            fprintf(stderr, "zeldovich_hip: uploading the RNG jump table failed\n");
            zd_plan_destroy(pl);
            return 1;
        }
        // z-walk of k_gen (see zd_device.h GenJumps): forward L rows inside a fold, then back to the
        // first term of the next k2; crossing z = N/2 shifts the counter row by +-(65536 - N)
        const long long wrap = 65536 - N, Lr = pl->L, back = -((long long) (R - 1) * Lr - 1);
        pl->J.fwd[0]  = row_jump(Lr);
        pl->J.fwd[1]  = row_jump(Lr + wrap);
        pl->J.back[0] = row_jump(back);
        pl->J.back[1] = row_jump(back > 0 ? back + wrap : back - wrap);
        pl->J.fwd_full[0]  = row_jump_full(Lr);
        pl->J.fwd_full[1]  = row_jump_full(Lr + wrap);
        pl->J.back_full[0] = row_jump_full(back);
        pl->J.back_full[1] = row_jump_full(back > 0 ? back + wrap : back - wrap);
        // the mirrored walker of genf_tile_kz: the same moves negated (crossing z = N/2 in its own direction of motion)
        const long long mf = -Lr, mb = -back;
        pl->J.mfwd[0]       = row_jump(mf);
        pl->J.mfwd[1]       = row_jump(mf - wrap);
        pl->J.mback[0]      = row_jump(mb);
        pl->J.mback[1]      = row_jump(mb > 0 ? mb + wrap : mb - wrap);
        pl->J.mfwd_full[0]  = row_jump_full(mf);
        pl->J.mfwd_full[1]  = row_jump_full(mf - wrap);
        pl->J.mback_full[0] = row_jump_full(mb);
        pl->J.mback_full[1] = row_jump_full(mb > 0 ? mb + wrap : mb - wrap);
    }
    {
        std::vector<cplx> twN = make_twiddles(pl->N), twL = make_twiddles(pl->L);
        PLCHECK(hipMalloc((void **) &pl->d_twN, sizeof(cplx) * twN.size()));
        PLCHECK(hipMemcpy(pl->d_twN, twN.data(), sizeof(cplx) * twN.size(), hipMemcpyHostToDevice));
        PLCHECK(hipMalloc((void **) &pl->d_twL, sizeof(cplx) * twL.size()));
        PLCHECK(hipMemcpy(pl->d_twL, twL.data(), sizeof(cplx) * twL.size(), hipMemcpyHostToDevice));
        if (np2) {  // composite transforms (zd_fft_q.h): exp(2 pi i k / P) | exp(2 pi i k / len) | exp(2 pi i k / Q) per length
            auto upload = [&](int len, cplx **dst) -> int {
                int P = 0, Q = 0;
                if (!zd::np2_split(len, &P, &Q)) return 1;
                std::vector<cplx> t = make_twiddles(P), b = make_twiddles(len), c = make_twiddles(Q);
                t.insert(t.end(), b.begin(), b.end());
                t.insert(t.end(), c.begin(), c.end());
                if (hipMalloc((void **) dst, sizeof(cplx) * t.size()) != hipSuccess) return 1;
                return hipMemcpy(*dst, t.data(), sizeof(cplx) * t.size(), hipMemcpyHostToDevice) != hipSuccess;
            };
            if (upload(pl->N, &pl->d_twq_n) || upload(pl->L, &pl->d_twq_l)) {
                fprintf(stderr, "zeldovich_hip: composite twiddle tables failed\n");
                zd_plan_destroy(pl);
                return 1;
            }
        }
        // LDS image of k_genf (layout: GenfTab in zd_kernels.hip).  Spline tables beyond 512 segments do not
        // fit: such runs use the general generator.
        const int nseg = pk->is_powerlaw ? 0 : pk->n - 1;
        if (nseg <= 512 && !tune_env("ZD_GEN_GENERAL")) {
            std::vector<double> T = build_genf_table(pk, nseg, &g.glut_x0, &g.glut_inv_dx);
            PLCHECK(hipMalloc((void **) &pl->d_genf, sizeof(double) * T.size()));
            PLCHECK(hipMemcpy(pl->d_genf, T.data(), sizeof(double) * T.size(), hipMemcpyHostToDevice));
            g.genf_tab  = pl->d_genf;
            g.genf_n    = (int) T.size();
            g.genf_nseg = nseg;
        }
    }
    // {P, 1/k^2} table over the integer |k|^2 that can carry power (zero rule of zeldovich.cpp:350-353)
    if (!tune_env("ZD_NO_PKTAB")) {
        const double half2 = (double) pl->half * pl->half;
        double nmax = p->corner_modes ? 3.0 * half2 : std::min(3.0 * half2, g.k2_cutoff / g.fundamental2 * (1 + 1e-9) + 2);
        const long long n = (long long) nmax + 2;
        if (n <= ((long long) 1 << 26)) {
            PLCHECK(hipMalloc((void **) &pl->d_pktab, sizeof(double) * 2 * (size_t) n));
            if (zd::launch_pk_table(g, (int) n, pl->d_pktab, 0)) {
                zd_plan_destroy(pl);
                return 1;
            }
            PLCHECK(hipDeviceSynchronize());
            g.pk_tab = (const double2 *) pl->d_pktab;
        }
    }
    if (phi_mode == 1 || phik) {  // M(k) = 2 g c^2 T(k) k^2 / (3 Omega_M H0^2), zeldovich.cpp:377-383
        const double H0 = 100., c = 299792.458, growth = 1. / (1 + p->z_initial);
        g.fnl_pre = 2. * growth * c * c;
        g.fnl_den = 3. * p->Omega_M * H0 * H0;
        g.n_s     = p->n_s;
        // Normalize(): primordial_norm = power(kmin) / primordial_power(kmin) with norm 1 (power_spectrum.cpp:221-222)
        g.primordial_norm = zd_pk_power(pk, pk->kmin) / (1. * exp(log(pk->kmin) * p->n_s));
        const long long n = 3LL * pl->half * pl->half + 2;
        PLCHECK(hipMalloc((void **) &pl->d_fnlM, sizeof(double) * (size_t) n));
        if (zd::launch_fnl_table(g, (int) n, pl->d_fnlM, 0)) {
            zd_plan_destroy(pl);
            return 1;
        }
        PLCHECK(hipDeviceSynchronize());
        g.fnl_M = pl->d_fnlM;
    }
    PLCHECK(hipMalloc((void **) &pl->d_red, sizeof(zd::Reduce)));
    PLCHECK(hipMemset(pl->d_red, 0, sizeof(zd::Reduce)));

    // ---- jobs of the z stage ----
    {
        zd::JobList &jl = pl->jobs;
        memset(&jl, 0, sizeof(jl));
        jl.pack = pl->pack;
        auto add = [&](int kind, int arr, int twin, int res) {
            jl.kind[jl.n] = kind;
            jl.arr[jl.n]  = arr;
            jl.twin[jl.n] = twin;
            jl.res[jl.n]  = res;
            jl.n++;
        };
        if (pl->pack == zd::PACK_PLTFIELD) {  // the six sums (k_zfft_f: job index = field index)
            add(zd::JOB_PX, 0, 0, 0);
            add(zd::JOB_PY, 1, 0, 0);
            add(zd::JOB_PZ, 2, 0, 0);
            add(zd::JOB_PFX, 3, 0, 0);
            add(zd::JOB_PFY, 4, 0, 0);
            add(zd::JOB_PFZ, 5, 0, 0);
        } else if (pl->pack == zd::PACK_ZAFIELD) {  // the potentials E, Z of the two residues (k_zfft_f: job index = field index)
            add(zd::JOB_E, 0, 0, 0);
            add(zd::JOB_Z, 1, 0, 0);
            add(zd::JOB_E, 2, 0, 1);
            add(zd::JOB_Z, 3, 0, 1);
            if (pl->dens) {  // ZD_qdensity = 1: the density sums of the two residues
                add(zd::JOB_DENS, 4, 0, 0);
                add(zd::JOB_DENS, 5, 0, 1);
            }
        } else if (pl->pack == zd::PACK_ZAPAIR) {  // (qy + i qz)_r0 | (qy + i qz)_r1 | qx_r0 + i qx_r1
            add(zd::JOB_B_SELF, 0, 0, 0);
            add(zd::JOB_B_TWIN, 0, 1, 0);
            add(zd::JOB_B_SELF, 1, 0, 1);
            add(zd::JOB_B_TWIN, 1, 1, 1);
            add(zd::JOB_FX, 2, 0, 0);  // generator output slot 4 = X_self, slot 5 = X_twin input (k_gen/k_genf combine F_x(r0), F_x(r1))
            add(zd::JOB_FX, 2, 1, 1);
        } else if (pl->pack == zd::PACK_PLT3) {  // qx + i vx | qy + i qz | vy + i vz
            add(zd::JOB_XV_SELF, 0, 0, 0);
            add(zd::JOB_XV_TWIN, 0, 1, 0);
            add(zd::JOB_B_SELF, 1, 0, 0);
            add(zd::JOB_B_TWIN, 1, 1, 0);
            add(zd::JOB_D_SELF, 2, 0, 0);
            add(zd::JOB_D_TWIN, 2, 1, 0);
        } else if (pl->narray == 1) {
            add(zd::JOB_DENS, 0, 0, 0);
        } else {
            add(zd::JOB_A_SELF, 0, 0, 0);
            add(zd::JOB_A_TWIN, 0, 1, 0);
            add(zd::JOB_B_SELF, 1, 0, 0);
            add(zd::JOB_B_TWIN, 1, 1, 0);
            if (pl->narray == 4) {
                add(zd::JOB_C_BOTH, 2, 0, 0);
                add(zd::JOB_D_SELF, 3, 0, 0);
                add(zd::JOB_D_TWIN, 3, 1, 0);
            }
        }
    }
    // ---- block store layout (zd_device.h StoreLayout): chunks per peer rank, ~2 MB tiles inside ----
    zd::StoreLayout &S = pl->S;
    S.N      = pl->N;
    S.half   = pl->half;
    S.Hq     = pl->Hq;
    S.lHq    = 0;
    while ((1 << S.lHq) < pl->Hq) S.lHq++;
    S.lG = 0;
    while ((1 << S.lG) < nranks) S.lG++;
    S.ky_stride = nranks;
    S.narray = pl->narray;
    S.prune     = tune_env("ZD_PRUNE") ? atoi(tune_env("ZD_PRUNE")) : 7;  // bit 0 k_gen, 1 k_zfft, 2 k_yfft
    S.nt = tune_env("ZD_NT") ? atoi(tune_env("ZD_NT")) : 0;
    if (phik) S.prune = 0;  // f_NL second pass: every mode carries power (the zero rule is bypassed)
    if (any_path) S.prune = 0;  // the convolution kernels transform every column: the generator must write every one
    S.kmax      = g.kmax;
    S.fund2     = g.fundamental2;
    S.k2_cutoff = p->corner_modes ? 0.0 : g.k2_cutoff;  // CornerModes: only the |k_i| == kmax rule prunes
    {
        const long long target = std::max<long long>(1, ((long long) 2 << 20) / ((long long) pl->N * 16));
        int lt = 0;
        while ((1LL << (lt + 1)) <= target) lt++;
        int lBk = (lt + 1) / 2, lBz = lt - lBk;
        S.rows_outer = 0;
        lBk = 20; lBz = 0;  // measured best on MI355X (profiles/r01_layout_sweep.txt): rows of one plane together
        if (const char *env = tune_env("ZD_LAYOUT")) {  // experimentation knob: "lBk,lBz,rows_outer"
            int a = lBk, b = lBz, o = 0;
            if (sscanf(env, "%d,%d,%d", &a, &b, &o) >= 2) {
                lBk = a; lBz = b; S.rows_outer = o;
            }
        }
        const int max_lBk = S.lHq + 1;  // self + twin slots of a chunk in ONE block: chunks are plane-major (zd_multi.cpp
                                        // exchanges them in plane groups)
        while (lBk > max_lBk) lBk--;
        while ((1 << lBz) > pl->Zq) lBz--;
        S.lBk = lBk;
        S.lBz = lBz;
        S.one_block = (nranks == 1 && lBk == S.lHq + 1 && lBz == 0 && !S.rows_outer) ? 1 : 0;
        int row_pad = store_row_pad(pl->N);  // in complex elements
        if (const char *env = tune_env("ZD_PAD")) sscanf(env, "%d", &row_pad);
        // Fused Z stage of the packed PLT store (zd_kernels_fz.hip: generator + z FFT in one kernel, the folded inputs stay on the
        // CU): one rank, z lines of 1024 or 512 points (PPD = 512 = BASELINE C2, 1024, or 2048 = BASELINE C3 at R = 2), the table generator's arithmetic,
        // version-2 streams, the whole field; and the kz = N/2 plane must be dead (the kernel never draws it: true unless
        // CornerModes is set together with ZD_k_cutoff != 1).  ZD_StoreMode = packed keeps the two-kernel stage (A/B runs).
        // Its store interleaves 4 planes along x (StoreLayout::lq = 2) so that a lane's 4 neighbours — 4 consecutive planes of
        // ONE column — write a 64-byte run.
        pl->fused_z = pl->pack == zd::PACK_PLT3 && p->store_mode == ZD_STORE_AUTO && S.one_block && zd::genz_plt_supported(pl->N, pl->L)
                      && g.genf_tab && !v1 && p->qoneslab < 0 && phi_mode == 0 && phik == nullptr && p->k_cutoff >= 1.0
                      && (g.kmax == pl->half || !p->corner_modes) && (S.prune & 7) == 7 && !tune_env("ZD_NO_FUSED_Z");
        S.lq         = pl->fused_z ? 2 : 0;
        // (the pad that spreads the y stage's strided rows over the HBM channels: 24 elements per plain row; per plane row of an
        // interleaved row 12 measured best — y stage of PPD=2048 PLT 144.5 ms at 24, 143.2 at 6, 139.7 at 12, 156.7 at 48)
        if (S.lq && !tune_env("ZD_PAD")) row_pad = store_row_pad(pl->N) / 2;
        S.pitch      = (pl->N + row_pad) << S.lq;
        S.a_rows     = (1 << lBk) << lBz;
        S.zb_rows    = S.a_rows * pl->narray;
        S.kb_rows    = S.zb_rows * ((pl->Zq >> S.lq) >> lBz);
        S.chunk_rows = S.kb_rows * ((2 * pl->Hq) >> lBk);
    }

    pl->ec.N        = pl->N;
    pl->ec.narray   = pl->narray;
    pl->ec.icformat = p->icformat;
    pl->ec.recsize  = record_size(p->icformat);
    pl->ec.qPLT     = p->qPLT;
    pl->ec.qdensity = p->qdensity;
    pl->ec.vnorm    = p->qPLT ? 1.0 : (sqrt(1. + 24 * p->f_cluster) - 1) * .25;  // output.cpp:78-82
    pl->ec.pack     = pl->pack == zd::PACK_PLTFIELD ? zd::PACK_PLT3 : pl->pack;  // the y stage builds the PLT3 arrays in the ring
    pl->ec.z_pair   = R / 2;
    pl->ec.xdead_lo = 1;  // (none; set below for the field stores' ring)
    pl->ec.xdead_hi = 0;
    pl->store_bytes_ = (int64_t) S.chunk_rows * S.pitch * nranks * 16;
    if (zd::pack_is_fields(pl->pack)) {
        std::vector<zd::FieldRow> rows;
        const int64_t fe = field_rows(p, nranks, &rows);
        PLCHECK(hipMalloc((void **) &pl->d_fieldrows, sizeof(zd::FieldRow) * rows.size()));
        PLCHECK(hipMemcpy(pl->d_fieldrows, rows.data(), sizeof(zd::FieldRow) * rows.size(), hipMemcpyHostToDevice));
        pl->F.lG          = S.lG;
        pl->F.lZq         = 0;
        while ((1 << pl->F.lZq) < pl->Zq) pl->F.lZq++;
        pl->F.Zq          = pl->Zq;
        pl->F.field_elems = fe;
        pl->F.ndens       = pl->dens ? 2 : 0;
        pl->F.nfield      = pl->pack == zd::PACK_PLTFIELD ? 6 : 4 + pl->F.ndens;
        pl->F.chunk_elems = (int64_t) pl->Zq * pl->F.nfield * fe;
        pl->F.rows        = pl->d_fieldrows;
        pl->store_bytes_  = pl->F.chunk_elems * nranks * 16;
        // ring: `ring_planes` store planes of the three arrays, laid out as a single-rank block store (one_block)
        pl->ring_planes = field_ring_planes(pl->N, pl->Zq);
        zd::StoreLayout &Q = pl->SR;
        Q            = S;
        Q.lq         = 0;
        Q.Hq         = pl->half;
        Q.lHq        = 0;
        while ((1 << Q.lHq) < pl->half) Q.lHq++;
        Q.lG         = 0;
        Q.ky_stride  = 1;
        Q.narray     = 3;
        Q.lBk        = Q.lHq + 1;
        Q.lBz        = 0;
        Q.rows_outer = 0;
        Q.one_block  = 1;
        Q.prune      = 0;
        Q.pitch      = pl->N + store_row_pad(pl->N);
        Q.a_rows     = pl->N;
        Q.zb_rows    = 3 * pl->N;
        Q.kb_rows    = Q.zb_rows * pl->ring_planes;
        Q.chunk_rows = Q.kb_rows;
        PLCHECK(zd_store_alloc((void **) &pl->d_ring, (size_t) Q.chunk_rows * Q.pitch * 16));
        if (pl->dens)  // the density array delta_r0 + i delta_r1 of the ring's planes: [plane][y][x]
            PLCHECK(zd_store_alloc((void **) &pl->d_ring_dens, (size_t) pl->ring_planes * pl->N * Q.pitch * 16));

    }
    // columns no row of which survives the zero rule (column_is_zero with ky = 0): never written by the z stage, skipped by the
    // y stage (whole tiles) and taken as zero by the x stage — power-of-two and composite kernels, every store; not the
    // convolution path (it transforms everything) and not the f_NL passes (prune = 0)
    if ((S.prune & 4) && !any_path && phi_mode == 0 && phik == nullptr) {
        int lo = 0;
        while (lo < pl->half && !zd::column_is_zero(S, lo, 0)) lo++;
        bool all = true;  // the rule is an interval in |kx| for every ky: check it rather than assume it
        for (int kx = lo; kx <= pl->half && all; kx++) all = zd::column_is_zero(S, kx, 0) && zd::column_is_zero(S, -kx, 0);
        if (all && lo >= 1) {
            pl->ec.xdead_lo = lo;
            pl->ec.xdead_hi = pl->N - lo;
            S.prune |= zd::PRUNE_YTILE;  // the x kernels take zeros there: the y stage may skip those tiles
        }
    }
    g.var_slots     = pl->d_red->sumsq;
    if (pl->fused_z) {
        // work items of k_genz_plt: for every half-space row ky >= 1 the live columns kx in (-w, w) (zero rule: column_is_zero), cut
        // where the draw counter of a row jumps (x = N/2 | N/2 + 1) and into pieces of at most 128 columns (a piece costs its
        // threads one jump from the row's stream head: ~14 steps of the 2^i table against 4 x 128 draws); longest first
        std::vector<zd::FzItem> items;
        const int SEG = 128;
        for (int kyl = (rank == 0 ? 1 : 0); kyl < pl->Hq; kyl++) {
            const int ky = rank + nranks * kyl;
            auto add_range = [&](int xa, int xb) {  // [xa, xb): maximal runs of live columns, in pieces
                int x = xa;
                while (x < xb) {
                    while (x < xb && zd::column_is_zero(S, x > pl->half ? x - pl->N : x, ky)) x++;
                    int e = x;
                    while (e < xb && !zd::column_is_zero(S, e > pl->half ? e - pl->N : e, ky)) e++;
                    for (int q = x; q < e; q += SEG) items.push_back(zd::FzItem{kyl, q, std::min(SEG, e - q), 0});
                    x = e;
                }
            };
            add_range(0, pl->half + 1);
            add_range(pl->half + 1, pl->N);
        }
        std::stable_sort(items.begin(), items.end(), [](const zd::FzItem &a, const zd::FzItem &b) { return a.n > b.n; });
        pl->n_fzitems = (unsigned) items.size();
        if (!items.empty()) {
            PLCHECK(hipMalloc((void **) &pl->d_fzitems, sizeof(zd::FzItem) * items.size()));
            PLCHECK(hipMemcpy(pl->d_fzitems, items.data(), sizeof(zd::FzItem) * items.size(), hipMemcpyHostToDevice));
        }
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&pl->ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (pl->ncu < 1) pl->ncu = 256;
    }

    // ---- folded-input slabs (two, for the gen||zfft overlap): enough rows per launch to fill the chip ----
    {
        const int64_t row_b = y_bytes_per_row(pl);
        int64_t slab_b = (int64_t) 3 << 29;  // ~1.5 GB per buffer
        if (const char *env = tune_env("ZD_SLAB_MB")) slab_b = (int64_t) atoll(env) << 20;
#ifdef ZD_SLAB_MB_FORCE  // experiment (make variant)
        slab_b = (int64_t) ZD_SLAB_MB_FORCE << 20;
#endif
        int rows = (int) std::max<int64_t>(1, slab_b / row_b);
        rows     = std::min(rows, pl->Hq);
        if (zd::pack_is_fields(pl->pack)) {  // whole row blocks
            rows = std::max(zd::FIELD_RB, rows / zd::FIELD_RB * zd::FIELD_RB);
            while (pl->Hq % rows) rows -= zd::FIELD_RB;
        }
        if (v1) {  // the accepted pairs of a slab: N^2 x 16 B per row, <= ~4 GB
            const int minr = zd::pack_is_fields(pl->pack) ? zd::FIELD_RB : 1;
            rows = std::max<int64_t>(minr, std::min<int64_t>(rows, ((int64_t) 4 << 30) / (N * N * 16)));
            if (zd::pack_is_fields(pl->pack)) rows = rows / zd::FIELD_RB * zd::FIELD_RB;
        }
        while (pl->Hq % rows) rows--;
        pl->slab_rows = rows;
        pl->overlap   = p->serial_z == 0 && !v1 && !any_path;  // version 1: the streams are sequential in ky
        // PLT with a table that is interpolated (eig_ppd does not divide into PPD): the (x, y) part of the trilinear lookup is
        // done once per column of a slab (k_eig_lines) instead of once per mode; one slab of lines, refilled on the generator's
        // stream in front of every generator launch
        if (g.qPLT && g.genf_tab && !v1 && eig_ppd % N != 0) {
            const size_t nb = (size_t) (eig_ppd / 2 + 1) * rows * N * 4 * sizeof(double);
            if (nb <= ((size_t) 1 << 30)) {
                PLCHECK(hipMalloc((void **) &pl->d_eiglines, nb));
                g.eig_lines = pl->d_eiglines;
                g.eig_rows  = rows;
            }
        }
        if (v1) {
            pl->v1_block = (int) (N / p->numblock);
            PLCHECK(hipMalloc((void **) &pl->d_v1streams, sizeof(zd::V1Stream) * (size_t) pl->v1_block));
            PLCHECK(hipMalloc((void **) &pl->d_v1dev, (size_t) rows * N * N * sizeof(double2)));
            PLCHECK(hipMalloc((void **) &pl->d_v1err, sizeof(int)));
            PLCHECK(hipMemset(pl->d_v1err, 0, sizeof(int)));
            g.v1dev = pl->d_v1dev;
        }
        // k_genf is a persistent kernel: a few workgroups per CU pull tiles from a counter (one per launch).  With
        // the second stream active the grid is kept small enough that a k_zfft workgroup (64 KB LDS, 2 waves/SIMD)
        // always fits beside the generator's waves on every CU.
        {
            int dev = 0, ncu = 256;
            hipGetDevice(&dev);
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
            int per_cu = pl->overlap ? 3 : 6;
            if (const char *env = tune_env("ZD_GEN_WGS")) per_cu = atoi(env);
#ifdef ZD_GEN_WGS_FORCE  // experiment (make variant)
            if (pl->overlap) per_cu = ZD_GEN_WGS_FORCE;
#endif
            pl->gen_max_wgs = std::max(1, per_cu) * std::max(1, ncu);
            pl->n_tilectr   = 2 * (pl->Hq / rows) + 2;  // per pass parity: the next pass's generator may already run
            PLCHECK(hipMalloc((void **) &pl->d_tilectr, sizeof(unsigned) * pl->n_tilectr));
        }
        // ring size: 1 slab without the overlap, else 2 + what the store (allocated by the caller afterwards) leaves
        int K = 1;
        if (pl->overlap) {
            K = 2;
            const int nslab = pl->Hq / rows;
            // Measured (PPD=4096 ZA, 60 slabs ahead): no gain — the generator's resident workgroups take LDS and
            // registers from k_yfft (128 KB LDS per workgroup at PPD=4096), which slows down by what the z stage gains.
            // The larger ring therefore stays an experiment: ZD_Y_AHEAD=1 sizes it from free HBM, ZD_Y_SLABS=n fixes it.
            if (tune_env("ZD_Y_AHEAD") && pl->npass > 1 && nslab > 2) {
                size_t free_b = 0, total_b = 0;
                hipMemGetInfo(&free_b, &total_b);
                const int64_t reserve = zd_plan_exchange_bytes(pl) * (nranks > 1 ? 2 : 1) + ((int64_t) 8 << 30);
                const int64_t spare   = (int64_t) free_b - reserve;
                if (spare > 0) K = (int) std::max<int64_t>(2, std::min<int64_t>(nslab, spare / (row_b * rows)));
            }
            if (const char *env = tune_env("ZD_Y_SLABS")) K = std::max(2, atoi(env));
        }
        pl->d_Y.assign(K, nullptr);
        for (int i = 0; i < K; i++) PLCHECK(hipMalloc((void **) &pl->d_Y[i], (size_t) row_b * rows));
        if (pl->overlap) {
#ifdef ZD_TUNING
            // A/B (VERDICT r4 #5 i): ZD_CU_SPLIT = n gives the z FFT's stream n CUs of every XCD's 32 and the generator's stream
            // the other 32 - n (hipExtStreamCreateWithCUMask; CU c of the mask = XCD c % 8, CU c / 8 of it on this part)
            if (const char *env = tune_env("ZD_CU_SPLIT")) {
                const int nf = std::max(1, std::min(31, atoi(env)));
                uint32_t mg[8] = {0}, mf[8] = {0};
                for (int c = 0; c < 256; c++) ((c / 8) < nf ? mf : mg)[c / 32] |= 1u << (c % 32);
                PLCHECK(hipExtStreamCreateWithCUMask(&pl->s_gen, 8, mg));
                PLCHECK(hipExtStreamCreateWithCUMask(&pl->s_fft, 8, mf));
            } else
#endif
            {
                PLCHECK(hipStreamCreateWithFlags(&pl->s_gen, hipStreamNonBlocking));
                PLCHECK(hipStreamCreateWithFlags(&pl->s_fft, hipStreamNonBlocking));
            }
            PLCHECK(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
            pl->ev_gen.assign(K, nullptr);
            pl->ev_fft.assign(K, nullptr);
            for (int i = 0; i < K; i++) {
                PLCHECK(hipEventCreateWithFlags(&pl->ev_gen[i], hipEventDisableTiming));
                PLCHECK(hipEventCreateWithFlags(&pl->ev_fft[i], hipEventDisableTiming));
            }
        }
    }
    if (any_path) {  // Bluestein tables for the z lines (length L) and the y / x lines (length N); simple [plane][array][y][x] store
        if (any_make_tab(pl->L, &pl->tabL, pl->d_any) || any_make_tab(pl->N, &pl->tabN, pl->d_any + 3)) {
            zd_plan_destroy(pl);
            return 1;
        }
        pl->AL.N         = pl->N;
        pl->AL.pitch     = pl->N + store_row_pad(pl->N);
        pl->AL.narray    = pl->narray;
        pl->store_bytes_ = (int64_t) pl->L * pl->narray * pl->N * pl->AL.pitch * 16;
    }
#undef PLCHECK
    *out = pl;
    return 0;
}

void zd_plan_destroy(zd_plan *pl) {
    if (!pl) return;
    collect_events(pl);
    hipFree(pl->d_fzitems);
    hipFree(pl->d_pk);
    hipFree(pl->d_lut);
    hipFree(pl->d_pktab);
    hipFree(pl->d_fnlM);
    hipFree(pl->d_eig);
    hipFree(pl->d_eiglines);
    hipFree(pl->d_rowstate);
    hipFree(pl->d_twN);
    hipFree(pl->d_twL);
    hipFree(pl->d_twq_n);
    hipFree(pl->d_twq_l);
    hipFree(pl->d_genf);
    hipFree(pl->d_tilectr);
    hipFree(pl->d_red);
    hipFree(pl->d_fieldrows);
    hipFree(pl->d_ring);
    hipFree(pl->d_ring_dens);
    hipFree(pl->d_dens_pass);
    zd_plan_destroy(pl->dens_sub);
    hipFree(pl->d_v1streams);
    hipFree(pl->d_v1dev);
    hipFree(pl->d_v1err);
    hipFree(pl->d_phik_owned);
    for (cplx *b : pl->d_any) hipFree(b);
    for (cplx *y : pl->d_Y) hipFree(y);
    if (pl->s_gen) hipStreamDestroy(pl->s_gen);
    if (pl->s_fft) hipStreamDestroy(pl->s_fft);
    if (pl->ev_fork) hipEventDestroy(pl->ev_fork);
    for (hipEvent_t e : pl->ev_pipe)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : pl->ev_gen)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : pl->ev_fft)
        if (e) hipEventDestroy(e);
    delete pl;
}

int32_t zd_plan_narray(const zd_plan *pl) { return pl->narray; }
int32_t zd_plan_store_mode(const zd_plan *pl) {
    return pl->pack == zd::PACK_NONE ? ZD_STORE_REFERENCE : zd::pack_is_fields(pl->pack) ? ZD_STORE_FIELDS : ZD_STORE_PACKED;
}
int32_t zd_plan_stream_factor(const zd_plan *pl) { return pl->R; }
int32_t zd_plan_record_size(const zd_plan *pl) { return pl->ec.recsize; }
int64_t zd_plan_exchange_bytes(const zd_plan *pl) { return pl->store_bytes_; }
int32_t zd_plan_passes(const zd_plan *pl) { return pl->npass; }
int32_t zd_plan_plane_step(const zd_plan *pl) { return pl->pstep; }
int64_t zd_plan_local_planes(const zd_plan *pl) { return (int64_t) pl->Zq * pl->pstep; }
int64_t zd_plan_plane_z(const zd_plan *pl, int pass, int64_t local_plane) {
    // PACK_ZAPAIR: store plane zl of pass p carries z-residues p (delivered plane 2 zl) and p + R/2 (plane 2 zl + 1)
    const int64_t zl = local_plane / pl->pstep, which = local_plane % pl->pstep;
    return (pass + which * (pl->R / 2)) + (int64_t) pl->R * ((int64_t) pl->rank * pl->Zq + zl);
}

// z FFT of one slab of generated rows into the block store (reference / packed arrays with Hermitian twins, or the
// potentials of the field store)
static int launch_zstage_fft(zd_plan *pl, int ky0, int kyloc0, int nky, const void *Y, void *d_send, hipStream_t st) {
    if (pl->d_twq_l) return zd::launch_zfft_fields_np2(pl->L, pl->F, pl->S, ky0, kyloc0, nky, Y, pl->d_twq_l, d_send, st);
    if (zd::pack_is_fields(pl->pack)) return zd::launch_zfft_fields(pl->L, pl->F, pl->S, ky0, kyloc0, nky, Y, pl->d_twL, d_send, st);
    return zd::launch_zfft(pl->L, pl->jobs, pl->S, ky0, kyloc0, nky, pl->Zq, Y, pl->d_twL, d_send, st);
}

// Z stage of the any-PPD path (zd_kernels_any.hip): general generator -> Y, convolution transform of Y along k2 in place,
// scatter into the [plane][array][y][x] store with the Hermitian twin rules
static int any_stage_z(zd_plan *pl, int residue, void *d_send, hipStream_t st) {
    const int zspan = span_begin(pl, ZD_K_ZSTAGE, st);
    if (pl->v1_block && zd::launch_v1_seed((unsigned long long) pl->p.seed, pl->v1_block, pl->d_v1streams, st)) return 1;
    HIPCHECK(hipMemsetAsync(pl->d_tilectr, 0, sizeof(unsigned) * pl->n_tilectr, st));
    int slab = 0;
    for (int r0 = 0; r0 < pl->Hq; r0 += pl->slab_rows, slab++) {
        const int nky = std::min(pl->slab_rows, pl->Hq - r0);
        tick(pl, ZD_K_GEN, st, true);
        if (pl->v1_block)  // ZD_Version = 1: rows that share a stream are drawn by successive launches (zd_plan_stage_z)
            for (int i0 = 0; i0 < nky; i0 += pl->v1_block)
                if (zd::launch_v1_draw(pl->g, pl->v1_block, r0 + i0, 1, std::min(pl->v1_block, nky - i0), pl->d_v1streams,
                                       pl->d_v1dev + (size_t) i0 * pl->N * pl->N, pl->d_v1err, st))
                    return 1;
        // (k_genf with walks of 16 / 4 / 2 z rows where the kernel exists, else the general generator)
        if (pl->d_eiglines && zd::launch_eig_lines(pl->g, r0, 1, nky, pl->d_eiglines, st)) return 1;
        if (zd::launch_gen(pl->g, pl->J, pl->jobs, pl->S, r0, nky, pl->L, residue, residue, pl->d_twN, pl->d_Y[0], pl->d_tilectr + slab,
                           pl->gen_max_wgs, st))
            return 1;
        tick(pl, ZD_K_GEN, st, false);
        tick(pl, ZD_K_ZFFT, st, true);
        if (zd::launch_any_cols(pl->tabL, pl->d_Y[0], (long long) pl->L * pl->N, pl->N, pl->N, pl->jobs.n * nky, -1, st)) return 1;
        if (zd::launch_any_scatter(pl->jobs, pl->AL, r0, nky, pl->L, pl->d_Y[0], d_send, st)) return 1;
        tick(pl, ZD_K_ZFFT, st, false);
    }
    span_end(pl, zspan, st);
    return 0;
}

static int stage_z_impl(zd_plan *pl, int residue, void *d_send, hipStream_t st, bool detached, hipEvent_t wait_ev, hipEvent_t done_ev);

// Z stage of a plan with the fused generator + z FFT (zd_kernels_fz.hip; one rank): the ky = 0 row — conjugate "loser" modes,
// zeldovich.cpp:485-503 — through the general generator and k_zfft (one row), every other row through k_genz_plt; one stream
static int fused_stage_z(zd_plan *pl, int residue, void *d_send, hipStream_t st, bool detached, hipEvent_t wait_ev, hipEvent_t done_ev) {
    if (detached && wait_ev) HIPCHECK(hipStreamWaitEvent(st, wait_ev, 0));
    pl->g.accum_var = pl->var_pending ? 1 : 0;  // once per run: every pass sees every mode
    pl->var_pending = false;
    const int zspan = span_begin(pl, ZD_K_ZSTAGE, st);
    HIPCHECK(hipMemsetAsync(pl->d_tilectr, 0, sizeof(unsigned), st));
    tick(pl, ZD_K_GEN, st, true);
    if (zd::launch_gen(pl->g, pl->J, pl->jobs, pl->S, 0, 1, pl->L, residue, residue, pl->d_twN, pl->d_Y[0], nullptr, pl->gen_max_wgs, st)) return 1;
    if (zd::launch_zfft(pl->L, pl->jobs, pl->S, 0, 0, 1, pl->Zq, pl->d_Y[0], pl->d_twL, d_send, st)) return 1;
    if (zd::launch_genz_plt(pl->g, pl->S, 0, pl->L, residue, pl->n_fzitems, pl->d_fzitems, pl->d_twN, pl->d_twL, d_send, pl->d_tilectr,
                            pl->ncu, st))
        return 1;
    tick(pl, ZD_K_GEN, st, false);
    span_end(pl, zspan, st);
    if (detached && done_ev) HIPCHECK(hipEventRecord(done_ev, st));
    return 0;
}

int zd_plan_stage_z(zd_plan *pl, int residue, void *d_send, void *hip_stream) {
    return stage_z_impl(pl, residue, d_send, (hipStream_t) hip_stream, false, nullptr, nullptr);
}

// The Z stage detached from the caller's stream (zd_multi.cpp pipelines passes with it): the z FFT — the only part that
// writes d_send — starts after `wait_ev` (NULL: at once) instead of after everything queued on `st`, nothing joins `st`, and
// `done_ev` is recorded behind the last z FFT.  Plans without the two worker streams (ZD_Version = 1, convolution PPDs,
// serial_z) run the stage on `st` as usual and record done_ev there.
int zd_plan_stage_z_detached(zd_plan *pl, int residue, void *d_send, void *hip_stream, void *wait_event, void *done_event) {
    return stage_z_impl(pl, residue, d_send, (hipStream_t) hip_stream, true, (hipEvent_t) wait_event, (hipEvent_t) done_event);
}

static int stage_z_impl(zd_plan *pl, int residue, void *d_send, hipStream_t st, bool detached, hipEvent_t wait_ev, hipEvent_t done_ev) {
    if (residue < 0 || residue >= pl->npass) return 1;
    if (pl->dens_sub) {
        // PLT + ZD_qdensity = 1 on a composite grid (plan_create_ex): first the whole density-only pass — Z stage into this store,
        // y / x stages of its density array — whose planes are the planes of this pass; then the PLT pass proper over the same
        // store.  All in stream order on st (a detached call included: the density planes are kept in ONE buffer).
        zd_plan *B = pl->dens_sub;
        if (detached && wait_ev) HIPCHECK(hipStreamWaitEvent(st, wait_ev, 0));
        B->pass_step = pl->pass_step;
        if (stage_z_impl(B, residue, d_send, st, false, nullptr, nullptr)) return 1;
        if (zd_plan_stage_x_group(B, residue, d_send, B->Zq, 0, 0, (int64_t) B->Zq * 2, nullptr, pl->d_dens_pass, st)) return 1;
        zd_plan *sub = pl->dens_sub;
        pl->dens_sub = nullptr;  // (the PLT pass itself)
        const int rc = stage_z_impl(pl, residue, d_send, st, false, nullptr, nullptr);
        pl->dens_sub = sub;
        if (rc) return rc;
        if (detached && done_ev) HIPCHECK(hipEventRecord(done_ev, st));
        return 0;
    }
    if (detached && (pl->any || !pl->overlap)) {  // no worker streams: in stream order on st
        if (wait_ev) HIPCHECK(hipStreamWaitEvent(st, wait_ev, 0));
        if (stage_z_impl(pl, residue, d_send, st, false, nullptr, nullptr)) return 1;
        if (done_ev) HIPCHECK(hipEventRecord(done_ev, st));
        return 0;
    }
    if (pl->any) return any_stage_z(pl, residue, d_send, st);
    if (pl->fused_z) return fused_stage_z(pl, residue, d_send, st, detached, wait_ev, done_ev);
    const int residue2 = pl->pstep == 2 ? residue + pl->R / 2 : residue;
    pl->g.accum_var = (pl->pack != zd::PACK_NONE && pl->var_pending) ? 1 : 0;  // once per run: every pass sees every mode
    pl->var_pending = false;
    const int ky_first = pl->rank, G = pl->nranks;  // this rank's half-space rows: rank, rank + G, ... (cyclic)
    const int zspan = span_begin(pl, ZD_K_ZSTAGE, detached ? pl->s_gen : st);
    if (!pl->overlap) {
        HIPCHECK(hipMemsetAsync(pl->d_tilectr, 0, sizeof(unsigned) * pl->n_tilectr, st));
        // ZD_Version = 1: every pass replays the streams from their seeds; rows that share a stream (block/G apart in
        // this rank's row order) are drawn by successive launches
        if (pl->v1_block && zd::launch_v1_seed((unsigned long long) pl->p.seed, pl->v1_block, pl->d_v1streams, st)) return 1;
        int slab = 0;
        for (int r0 = 0; r0 < pl->Hq; r0 += pl->slab_rows, slab++) {
            const int nky = std::min(pl->slab_rows, pl->Hq - r0);
            tick(pl, ZD_K_GEN, st, true);
            if (pl->v1_block) {
                const int group = pl->v1_block / G;
                for (int i0 = 0; i0 < nky; i0 += group)
                    if (zd::launch_v1_draw(pl->g, pl->v1_block, ky_first + G * (r0 + i0), G, std::min(group, nky - i0), pl->d_v1streams,
                                           pl->d_v1dev + (size_t) i0 * pl->N * pl->N, pl->d_v1err, st))
                        return 1;
            }
            if (pl->d_eiglines && zd::launch_eig_lines(pl->g, ky_first + G * r0, G, nky, pl->d_eiglines, st)) return 1;
            if (zd::launch_gen(pl->g, pl->J, pl->jobs, pl->S, ky_first + G * r0, nky, pl->L, residue, residue2, pl->d_twN, pl->d_Y[0],
                               pl->d_tilectr + slab, pl->gen_max_wgs, st))
                return 1;
            tick(pl, ZD_K_GEN, st, false);
            tick(pl, ZD_K_ZFFT, st, true);
            if (launch_zstage_fft(pl, ky_first + G * r0, r0, nky, pl->d_Y[0], d_send, st)) return 1;
            tick(pl, ZD_K_ZFFT, st, false);
        }
        span_end(pl, zspan, st);
        return 0;
    }
    // Two worker streams.  s_fft (k_zfft, writes the store) starts after everything already queued on the caller's
    // stream; s_gen (generator, writes ring slots only) is ordered by the ring alone and may be a pass ahead.
    const int K = (int) pl->d_Y.size(), nslab = pl->Hq / pl->slab_rows;
    if (!detached) {
        HIPCHECK(hipEventRecord(pl->ev_fork, st));
        HIPCHECK(hipStreamWaitEvent(pl->s_fft, pl->ev_fork, 0));
    } else if (wait_ev) {
        HIPCHECK(hipStreamWaitEvent(pl->s_fft, wait_ev, 0));
    }
    auto issue_gen = [&](int pass, int slab, long long gno, int accum) -> int {
        const int slot = (int) (gno % K), r0 = slab * pl->slab_rows;
        // the slot is free once the k_zfft that read its previous content (slab gno - K) has finished
        if (gno >= K) HIPCHECK(hipStreamWaitEvent(pl->s_gen, pl->ev_fft[slot], 0));
        unsigned *ctr = pl->d_tilectr + (pass & 1) * nslab;
        if (slab == 0) HIPCHECK(hipMemsetAsync(ctr, 0, sizeof(unsigned) * nslab, pl->s_gen));
        pl->g.accum_var = accum;
        tick(pl, ZD_K_GEN, pl->s_gen, true);
        if (pl->d_eiglines && zd::launch_eig_lines(pl->g, ky_first + G * r0, G, pl->slab_rows, pl->d_eiglines, pl->s_gen)) return 1;
        if (zd::launch_gen(pl->g, pl->J, pl->jobs, pl->S, ky_first + G * r0, pl->slab_rows, pl->L, pass,
                           pl->pstep == 2 ? pass + pl->R / 2 : pass, pl->d_twN, pl->d_Y[slot], ctr + slab,
                           pl->gen_max_wgs, pl->s_gen))
            return 1;
        tick(pl, ZD_K_GEN, pl->s_gen, false);
        HIPCHECK(hipEventRecord(pl->ev_gen[slot], pl->s_gen));
        return 0;
    };
    const int accum = pl->g.accum_var;
    if (pl->ahead_pass != residue) pl->ahead_n = 0;  // nothing (or something else) was started ahead
    for (int slab = 0; slab < nslab; slab++) {
        long long gno;
        if (slab < pl->ahead_n)
            gno = pl->ahead_g0 + slab;
        else {
            gno = pl->next_g++;
            if (issue_gen(residue, slab, gno, accum)) return 1;
        }
        const int slot = (int) (gno % K), r0 = slab * pl->slab_rows;
        HIPCHECK(hipStreamWaitEvent(pl->s_fft, pl->ev_gen[slot], 0));
        tick(pl, ZD_K_ZFFT, pl->s_fft, true);
        if (launch_zstage_fft(pl, ky_first + G * r0, r0, pl->slab_rows, pl->d_Y[slot], d_send, pl->s_fft)) return 1;
        tick(pl, ZD_K_ZFFT, pl->s_fft, false);
        HIPCHECK(hipEventRecord(pl->ev_fft[slot], pl->s_fft));
    }
    pl->ahead_pass = -1;
    pl->ahead_n    = 0;
    // run ahead: the first slabs of the next pass go into the ring now; they execute while the caller's stream does
    // this pass's y and x transforms
    if (K > 2 && residue + pl->pass_step < pl->npass && p_oneslab_off(pl)) {
        const int n = std::min(nslab, K);
        pl->ahead_pass = residue + pl->pass_step;  // (this rank's next pass: with pass groups not residue + 1)
        pl->ahead_g0   = pl->next_g;
        for (int slab = 0; slab < n; slab++)
            if (issue_gen(residue + pl->pass_step, slab, pl->next_g++, 0)) return 1;
        pl->ahead_n = n;
    }
    if (detached) {  // nothing joins the caller's stream: whoever needs the store waits for done_ev
        if (done_ev) HIPCHECK(hipEventRecord(done_ev, pl->s_fft));
        span_end(pl, zspan, pl->s_fft);
        return 0;
    }
    // join: the caller's stream continues after the last k_zfft of this pass
    HIPCHECK(hipEventRecord(pl->ev_fork, pl->s_fft));
    HIPCHECK(hipStreamWaitEvent(st, pl->ev_fork, 0));
    span_end(pl, zspan, st);
    return 0;
}

// The XY stages address the store through its layout; after an exchange in plane groups (zd_multi.cpp) the data sits in
// a ring slot whose chunks are `chunk_planes` planes long instead of Zq.  Chunks are plane-major, so only the chunk stride
// changes.
static zd::StoreLayout layout_for_chunks(const zd_plan *pl, int chunk_planes) {
    zd::StoreLayout S = pl->S;
    if (chunk_planes != pl->Zq) {
        S.kb_rows    = S.zb_rows * (chunk_planes >> S.lBz);
        S.chunk_rows = S.kb_rows * ((2 * pl->Hq) >> S.lBk);
    }
    return S;
}
static zd::FieldLayout fields_for_chunks(const zd_plan *pl, int chunk_planes) {
    zd::FieldLayout F = pl->F;
    F.chunk_elems     = (long long) chunk_planes * F.nfield * F.field_elems;
    return F;
}

// y stage on a store (or ring slot) whose chunks hold `chunk_planes` planes, for its planes [0, nplanes)
int zd_plan_stage_y_group(zd_plan *pl, void *d_recv, int chunk_planes, int nplanes, void *hip_stream) {
    hipStream_t st = (hipStream_t) hip_stream;
    if (zd::pack_is_fields(pl->pack)) return 0;  // field stores: the y transform runs plane group by plane group in stage_x
    if (pl->any) {  // every (plane, array) image: columns along y, the Nyquist row counted as zero (zeldovich.cpp:644-650)
        tick(pl, ZD_K_YFFT, st, true);
        if (zd::launch_any_cols(pl->tabN, d_recv, (long long) pl->N * pl->AL.pitch, pl->AL.pitch, pl->N, nplanes * pl->narray, pl->N / 2, st))
            return 1;
        tick(pl, ZD_K_YFFT, st, false);
        return 0;
    }
    tick(pl, ZD_K_YFFT, st, true);
    if (zd::launch_yfft(layout_for_chunks(pl, chunk_planes), nplanes, pl->d_twN, d_recv, st)) return 1;
    tick(pl, ZD_K_YFFT, st, false);
    return 0;
}
int zd_plan_stage_y(zd_plan *pl, void *d_recv, void *hip_stream) { return zd_plan_stage_y_group(pl, d_recv, pl->Zq, pl->Zq, hip_stream); }

// x stage (+ the y stage of the field store) for delivered planes [plane0, plane0 + nplanes) of a store / ring slot with
// `chunk_planes` planes per chunk; `gplane0` = the local plane index of the first one inside the pass (it fixes z)
int zd_plan_stage_x_group(zd_plan *pl, int residue, const void *d_recv, int chunk_planes, int64_t plane0, int64_t gplane0,
                          int64_t nplanes, void *d_records, float *d_density, void *hip_stream) {
    hipStream_t st = (hipStream_t) hip_stream;
    const int ps = pl->pstep;
    if (plane0 < 0 || nplanes < 1 || plane0 + nplanes > (int64_t) chunk_planes * ps || plane0 % ps || nplanes % ps || gplane0 % ps
        || gplane0 + nplanes > (int64_t) pl->Zq * ps) {
        fprintf(stderr, "zeldovich_hip: stage_x plane range [%lld, +%lld) invalid (multiples of %d inside [0, %lld))\n",
                (long long) plane0, (long long) nplanes, ps, (long long) chunk_planes * ps);
        return 1;
    }
    if (d_density && pl->dens_sub) {  // the planes the density-only half left behind at the head of this pass's Z stage
        HIPCHECK(hipMemcpyAsync(d_density, pl->d_dens_pass + (size_t) gplane0 * pl->N * pl->N, (size_t) nplanes * pl->N * pl->N * sizeof(float),
                                hipMemcpyDeviceToDevice, st));
        d_density = nullptr;
    }
    if (d_density && pl->pack != zd::PACK_NONE && !pl->dens) return 1;  // packed stores carry no density field (but the six-field store)
    if (pl->any) {  // x lines of the planes in place (each plane once), then the particle epilogue
        const int z_first = (int) zd_plan_plane_z(pl, residue, gplane0);
        cplx *first = (cplx *) const_cast<void *>(d_recv) + (long long) plane0 * pl->narray * pl->N * pl->AL.pitch;
        tick(pl, ZD_K_XFFT, st, true);
        if (zd::launch_any_lines(pl->tabN, first, pl->AL.pitch, (long long) nplanes * pl->narray * pl->N, st)) return 1;
        if (zd::launch_any_emit(pl->AL, pl->ec, d_recv, (int) plane0, (int) nplanes, z_first, pl->R, d_records, d_density, pl->d_red, st)) return 1;
        tick(pl, ZD_K_XFFT, st, false);
        return 0;
    }
    if (zd::pack_is_fields(pl->pack)) {
        // y stage (potentials -> the three displacement arrays of a group of store planes, into the ring) + x stage
        const zd::FieldLayout F = fields_for_chunks(pl, chunk_planes);
        const int p0 = (int) (plane0 / ps), np = (int) (nplanes / ps);
        for (int g0 = 0; g0 < np; g0 += pl->ring_planes) {
            const int ng = std::min(pl->ring_planes, np - g0);
            char *rec_g = d_records ? (char *) d_records + (size_t) g0 * ps * pl->N * pl->N * pl->ec.recsize : nullptr;
            const int z_first = (int) zd_plan_plane_z(pl, residue, gplane0 + (int64_t) g0 * ps);
            tick(pl, ZD_K_YFFT, st, true);
            if (!pl->dens_only
                && (pl->d_twq_n ? zd::launch_yfft_fields_np2(F, pl->S, pl->d_twq_n, d_recv, p0 + g0, ng, pl->SR.pitch, pl->d_ring, 0, st)
                                : zd::launch_yfft_fields(F, pl->S, pl->d_twN, d_recv, p0 + g0, ng, pl->SR.pitch, pl->d_ring, st)))
                return 1;
            if (pl->dens && d_density
                && zd::launch_yfft_fields_np2(F, pl->S, pl->d_twq_n, d_recv, p0 + g0, ng, pl->SR.pitch, pl->d_ring_dens, 1, st))
                return 1;
            tick(pl, ZD_K_YFFT, st, false);
            tick(pl, ZD_K_XFFT, st, true);
            if (!pl->dens_only
                && (pl->d_twq_n ? zd::launch_xfft_np2(pl->N, pl->ec, pl->d_twq_n, pl->d_ring, pl->SR.pitch, ng, z_first, pl->R, rec_g, pl->d_red, st)
                                : zd::launch_xfft(pl->SR, pl->ec, pl->d_twN, pl->d_ring, 0, ng, z_first, pl->R, rec_g, nullptr, pl->d_red, st)))
                return 1;
            if (pl->dens && d_density
                && zd::launch_xdens_np2(pl->N, pl->d_twq_n, pl->d_ring_dens, pl->SR.pitch, ng, d_density + (size_t) g0 * ps * pl->N * pl->N, st))
                return 1;
            tick(pl, ZD_K_XFFT, st, false);
        }
        return 0;
    }
    const int z_first = (int) zd_plan_plane_z(pl, residue, gplane0);
    tick(pl, ZD_K_XFFT, st, true);
    if (zd::launch_xfft(layout_for_chunks(pl, chunk_planes), pl->ec, pl->d_twN, d_recv, (int) (plane0 / ps), (int) (nplanes / ps), z_first,
                        pl->R, d_records, d_density, pl->d_red, st))
        return 1;
    tick(pl, ZD_K_XFFT, st, false);
    return 0;
}
int zd_plan_stage_x(zd_plan *pl, int residue, const void *d_recv, int64_t plane0, int64_t nplanes, void *d_records,
                    float *d_density, void *hip_stream) {
    return zd_plan_stage_x_group(pl, residue, d_recv, pl->Zq, plane0, plane0, nplanes, d_records, d_density, hip_stream);
}

void zd_plan_tick(zd_plan *pl, int kind, void *hip_stream, int begin) { tick(pl, kind, (hipStream_t) hip_stream, begin != 0); }

int zd_plan_stats(zd_plan *pl, zd_stats *out) {
    HIPCHECK(hipDeviceSynchronize());
    collect_events(pl);
    zd::Reduce h;
    HIPCHECK(hipMemcpy(&h, pl->d_red, sizeof(h), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemset(pl->d_red, 0, sizeof(zd::Reduce)));
    memset(out, 0, sizeof(*out));
    double ss = 0;
    for (int i = 0; i < zd::NSLOT; i++) ss += h.sumsq[i];
    // packed stores: ss = sum over the full k cube of |D(k)|^2 (generator); sum_x delta^2 = N^3 sum_k |D|^2
    out->density_variance = pl->pack != zd::PACK_NONE ? ss * (double) pl->N * (double) pl->N * (double) pl->N : ss;
    pl->var_pending       = true;
    for (int j = 0; j < 3; j++) {
        // output.cpp:190-193 keeps the signed value of the largest |pos|, the first one in (z, y, x) order on a tie: the slots hold
        // (|v|, (linear index << 1) | negative) of their best record
        unsigned long long babs = 0, bkey = 0;
        for (int i = 0; i < zd::NSLOT; i++)
            if (h.maxabs[j][i] > babs || (h.maxabs[j][i] == babs && babs != 0 && h.maxkey[j][i] < bkey)) {
                babs = h.maxabs[j][i];
                bkey = h.maxkey[j][i];
            }
        double a;
        memcpy(&a, &babs, 8);
        out->max_disp[j]       = (bkey & 1ULL) ? -a : a;
        out->max_disp_index[j] = babs ? (int64_t) (bkey >> 1) : -1;
    }
    for (int k = 0; k < ZD_K_COUNT; k++) {
        out->kernel_ms[k]       = pl->kernel_ms[k];
        out->kernel_launches[k] = pl->launches[k];
        pl->kernel_ms[k]        = 0;
        pl->launches[k]         = 0;
    }
    if (pl->dens_sub) {  // the density-only half of a PLT + ZD_qdensity = 1 run: its kernels belong to this run's time
        zd_stats sb;
        if (zd_plan_stats(pl->dens_sub, &sb)) return 1;
        for (int k = 0; k < ZD_K_COUNT; k++) {
            out->kernel_ms[k] += sb.kernel_ms[k];
            out->kernel_launches[k] += sb.kernel_launches[k];
        }
    }
    out->bytes_intermediate = zd_plan_exchange_bytes(pl);
    out->bytes_sent         = pl->bytes_sent;
    pl->bytes_sent          = 0;
    out->stream_factor      = pl->R;
    out->modes_cached       = 0;  // modes are regenerated per residue pass (counter-addressed RNG)
    if (pl->d_v1err) {  // ZD_Version = 1: a stream that stopped accepting pairs (k_v1_draw's guard)
        int e = 0;
        HIPCHECK(hipMemcpy(&e, pl->d_v1err, sizeof(int), hipMemcpyDeviceToHost));
        if (e) {
            fprintf(stderr, "zeldovich_hip: ZD_Version = 1 stream generator failed\n");
            return 1;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
int zd_generate(const zd_params *p_in, const zd_pk *pk, const double *eig, int64_t eig_ppd, zd_slab_cb cb,
                void *user, zd_stats *out) {
    // ZD_NumGPU: one host thread per GPU, exchange inside the library (zd_multi.cpp).  (ZD_qoneslab finishes ONE plane of one
    // pass and reports the reductions of that slab alone, output.cpp:197: that is this single-GPU path's job.)
    if (p_in->ngpu > 1 && p_in->qoneslab < 0) {
        int ndev = 0;
        HIPCHECK(hipGetDeviceCount(&ndev));
        // RCCL between distinct GPUs; when the ranks have to share devices (fewer GPUs than ranks: test boxes) they
        // pull their slices with device copies instead
        return zd_generate_multi(p_in, pk, eig, eig_ppd, cb, user, out, ndev >= p_in->ngpu ? 0 : 1);
    }
    zd_params p = *p_in;
    const int64_t N = p.ppd;
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    if (p.stream_factor <= 0) {
        // leave 16 GB for tables, the folded-input slabs (~3 GB), the record ring (8 GB) and the runtime
        const int64_t budget = (int64_t) free_b - ((int64_t) 16 << 30);
        const int R = zd_choose_stream_factor(&p, 1, budget);
        if (R < 0) {
            fprintf(stderr, "zeldovich_hip: PPD %lld does not fit in %.1f GB of free HBM at any stream factor\n",
                    (long long) N, free_b / 1e9);
            return 1;
        }
        p.stream_factor = R;
    }
    // ---- f_NL: phi field -> local transform -> Fourier space again (zeldovich.cpp:945-960) ----
    cplx *d_phik = nullptr;
    if (p.f_NL != 0.) {
        if (make_phik(&p, pk, &d_phik)) return 1;
        HIPCHECK(hipMemGetInfo(&free_b, &total_b));
        if (p_in->stream_factor <= 0) {
            const int R2 = zd_choose_stream_factor(&p, 1, (int64_t) free_b - ((int64_t) 16 << 30));
            if (R2 < 0) {
                hipFree(d_phik);
                return 1;
            }
            p.stream_factor = R2;
        }
    }
    zd_plan *pl = nullptr;
    if (plan_create_ex(&p, pk, eig, eig_ppd, 0, 1, 0, d_phik, &pl)) {
        hipFree(d_phik);
        return 1;
    }
    const int R = pl->R, recsize = pl->ec.recsize, npass = pl->npass, pstep = pl->pstep;
    const int64_t Pp = zd_plan_local_planes(pl);  // planes delivered per pass
    const bool want_rec = pl->narray >= 2 && !pl->dens_only, want_dens = p.qdensity != 0;

    // Delivery (WriteParticlesSlab's role, src/output.cpp:207-224) is asynchronous: NB = 2 record buffers on the device
    // and in pinned host memory; chunk i is produced into buffer i % 2 on the compute stream, copied D2H on a copy stream
    // and handed to a writer thread that runs the callback plane by plane — the x stage of chunk i+1, the PCIe copy of
    // chunk i and the callback of chunk i-1 overlap (round 1 serialised the three).  The callback is still invoked
    // from ONE thread, in production order.
    constexpr int NB = 2;
    void *d_store = nullptr, *d_rec[NB] = {nullptr, nullptr};
    float *d_dens[NB] = {nullptr, nullptr};
    char *h_rec[NB] = {nullptr, nullptr};
    float *h_dens[NB] = {nullptr, nullptr};
    hipEvent_t ev_x[NB] = {nullptr, nullptr}, ev_c[NB] = {nullptr, nullptr};
    hipStream_t s_copy = nullptr;
    int rc = 1;
    hipStream_t st = 0;
    // planes handed over per x-stage launch (a multiple of the plane step)
    const int64_t plane_b = N * N * (int64_t) (want_rec ? recsize : 0) + (want_dens ? N * N * 4 : 0);
    // (with a host callback each buffer is mirrored in pinned host memory: 1 GB; the NULL sink takes 8 GB so that an
    // x-stage launch covers several planes even at PPD=4096, where one plane of records is 537 MB)
    const int64_t ring_b = cb ? ((int64_t) 1 << 30) : ((int64_t) 8 << 30);
    int chunk = (int) std::max<int64_t>(1, std::min<int64_t>(Pp, ring_b / std::max<int64_t>(plane_b, 1)));
    chunk     = std::max(pstep, chunk / pstep * pstep);
    const int nbuf = cb ? NB : 1;
    struct Item {
        int slot, pass;
        int64_t first, n, only;  // only >= 0: deliver just that local plane (ZD_qoneslab)
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> queue;
    bool slot_busy[NB] = {false, false}, done = false;
    std::atomic<int> cb_failed{0};
    std::thread writer;
    std::chrono::steady_clock::time_point t0;
    do {
        if (zd_store_alloc(&d_store, (size_t) zd_plan_exchange_bytes(pl)) != hipSuccess) {
            fprintf(stderr, "zeldovich_hip: cannot allocate the %.2f GB block store\n", zd_plan_exchange_bytes(pl) / 1e9);
            break;
        }
        bool ok = true;
        for (int b2 = 0; b2 < nbuf && ok; b2++) {
            if (want_rec && hipMalloc(&d_rec[b2], (size_t) chunk * N * N * recsize) != hipSuccess) ok = false;
            if (want_dens && hipMalloc((void **) &d_dens[b2], (size_t) chunk * N * N * 4) != hipSuccess) ok = false;
            if (cb) {
                if (want_rec && hipHostMalloc((void **) &h_rec[b2], (size_t) chunk * N * N * recsize) != hipSuccess) ok = false;
                if (want_dens && hipHostMalloc((void **) &h_dens[b2], (size_t) chunk * N * N * 4) != hipSuccess) ok = false;
                if (hipEventCreateWithFlags(&ev_x[b2], hipEventDisableTiming) != hipSuccess) ok = false;
                if (hipEventCreateWithFlags(&ev_c[b2], hipEventDisableTiming) != hipSuccess) ok = false;
            }
        }
        if (!ok) break;
        if (cb && hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking) != hipSuccess) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (cb) {
            writer = std::thread([&]() {
                for (;;) {
                    Item it;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return done || !queue.empty(); });
                        if (queue.empty()) return;
                        it = queue.front();
                        queue.pop_front();
                    }
                    if (hipEventSynchronize(ev_c[it.slot]) != hipSuccess) cb_failed.store(1);
                    for (int64_t i = 0; i < it.n && !cb_failed.load(); i++) {
                        if (it.only >= 0 && it.first + i != it.only) continue;
                        const int64_t z = zd_plan_plane_z(pl, it.pass, it.first + i);
                        if (cb(user, z, N * N, want_rec ? h_rec[it.slot] + (size_t) i * N * N * recsize : nullptr,
                               want_dens ? h_dens[it.slot] + (size_t) i * N * N : nullptr))
                            cb_failed.store(1);
                    }
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        slot_busy[it.slot] = false;
                    }
                    cv.notify_all();
                }
            });
        }
        t0 = std::chrono::steady_clock::now();
        bool fail = false;
        // ZD_qoneslab (zeldovich.cpp:669): only that slab is wanted -> only its pass and its store plane are finished
        int want_pass = -1;
        int64_t want_plane = -1;
        if (p.qoneslab >= 0) {
            const int r = p.qoneslab % R;
            want_pass   = r % npass;
            want_plane  = (int64_t) (p.qoneslab / R) * pstep + r / npass;
        }
        long long nchunk = 0;
        for (int r = 0; r < npass && !fail; r++) {
            if (want_pass >= 0 && r != want_pass) continue;
            if (zd_plan_stage_z(pl, r, d_store, st) || zd_plan_stage_y(pl, d_store, st)) {
                fail = true;
                break;
            }
            for (int64_t pl0 = 0; pl0 < Pp && !fail; pl0 += chunk) {
                int64_t first = pl0, n = std::min<int64_t>(chunk, Pp - pl0);
                if (want_plane >= 0) {
                    if (want_plane < pl0 || want_plane >= pl0 + n) continue;
                    first = want_plane / pstep * pstep;
                    n     = pstep;
                }
                const int slot = cb ? (int) (nchunk++ % NB) : 0;
                if (cb) {  // the writer has finished with this slot's host buffer (chunk i - 2) — and with it its D2H copy
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !slot_busy[slot] || cb_failed.load(); });
                    if (cb_failed.load()) {
                        fail = true;
                        break;
                    }
                    slot_busy[slot] = true;
                }
                if (zd_plan_stage_x(pl, r, d_store, first, n, d_rec[slot], d_dens[slot], st)) {
                    fail = true;
                    break;
                }
                if (cb) {
                    if (hipEventRecord(ev_x[slot], st) != hipSuccess || hipStreamWaitEvent(s_copy, ev_x[slot], 0) != hipSuccess) fail = true;
                    if (want_rec && hipMemcpyAsync(h_rec[slot], d_rec[slot], (size_t) n * N * N * recsize, hipMemcpyDeviceToHost, s_copy) != hipSuccess) fail = true;
                    if (want_dens && hipMemcpyAsync(h_dens[slot], d_dens[slot], (size_t) n * N * N * 4, hipMemcpyDeviceToHost, s_copy) != hipSuccess) fail = true;
                    if (hipEventRecord(ev_c[slot], s_copy) != hipSuccess) fail = true;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        queue.push_back(Item{slot, r, first, n, want_plane});
                    }
                    cv.notify_all();
                }
            }
        }
        if (cb) {  // drain
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return (queue.empty() && !slot_busy[0] && !slot_busy[1]) || cb_failed.load(); });
            }
            if (cb_failed.load()) fail = true;
        }
        if (fail) break;
        if (zd_plan_stats(pl, out)) break;
        out->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rc = 0;
    } while (0);
    if (writer.joinable()) {
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
            queue.clear();
        }
        cv.notify_all();
        writer.join();
    }
    hipDeviceSynchronize();
    hipFree(d_phik);
    hipFree(d_store);
    for (int b2 = 0; b2 < NB; b2++) {
        hipFree(d_rec[b2]);
        hipFree(d_dens[b2]);
        if (h_rec[b2]) hipHostFree(h_rec[b2]);
        if (h_dens[b2]) hipHostFree(h_dens[b2]);
        if (ev_x[b2]) hipEventDestroy(ev_x[b2]);
        if (ev_c[b2]) hipEventDestroy(ev_c[b2]);
    }
    if (s_copy) hipStreamDestroy(s_copy);
    zd_plan_destroy(pl);
    return rc;
}

#ifdef ZD_TESTING
// ------------------------------------------------------------------------------------------------
// device test hooks

static int make_test_gen(const zd_params *p, const zd_pk *pk, zd_plan **pl) {
    zd_params q = *p;
    q.qPLT      = 0;
    q.stream_factor = q.ppd > 4096 ? (int) (q.ppd / 4096) : 1;  // z-FFT kernels exist up to length 4096
    return zd_plan_create(&q, pk, nullptr, 0, 0, 1, pl);
}

int zd_test_draws(int64_t seed, int64_t n, const int32_t *kxyz, uint64_t *out) {
    // only the RNG members of GenConst are used by the draw path
    zd_params p;
    memset(&p, 0, sizeof(p));
    p.ppd = 4096;  // row table covers ky < 2048
    p.seed = seed;
    p.boxsize = 1;
    p.fundamental = 1;
    p.nyquist = 1;
    p.k_cutoff = 1;
    p.f_cluster = 1;
    zd_pk pk;
    memset(&pk, 0, sizeof(pk));
    pk.is_powerlaw = 1;
    pk.powerlaw_index = -1;
    pk.normalization = 1;
    zd_plan *pl = nullptr;
    if (make_test_gen(&p, &pk, &pl)) return 1;
    int *d_k = nullptr;
    uint64_t *d_o = nullptr;
    int rc = 1;
    do {
        if (hipMalloc((void **) &d_k, sizeof(int) * 3 * n) != hipSuccess) break;
        if (hipMalloc((void **) &d_o, sizeof(uint64_t) * 2 * n) != hipSuccess) break;
        if (hipMemcpy(d_k, kxyz, sizeof(int) * 3 * n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (zd::launch_test_modes(pl->g, n, d_k, d_o, nullptr, 0)) break;
        if (hipMemcpy(out, d_o, sizeof(uint64_t) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    hipFree(d_k);
    hipFree(d_o);
    zd_plan_destroy(pl);
    return rc;
}

int zd_test_modes(const zd_params *p, const zd_pk *pk, int64_t n, const int32_t *kxyz, double *D) {
    zd_plan *pl = nullptr;
    if (make_test_gen(p, pk, &pl)) return 1;
    int *d_k = nullptr;
    double *d_o = nullptr;
    int rc = 1;
    do {
        if (hipMalloc((void **) &d_k, sizeof(int) * 3 * n) != hipSuccess) break;
        if (hipMalloc((void **) &d_o, sizeof(double) * 2 * n) != hipSuccess) break;
        if (hipMemcpy(d_k, kxyz, sizeof(int) * 3 * n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (zd::launch_test_modes(pl->g, n, d_k, nullptr, d_o, 0)) break;
        if (hipMemcpy(D, d_o, sizeof(double) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    hipFree(d_k);
    hipFree(d_o);
    zd_plan_destroy(pl);
    return rc;
}

int zd_test_modes_table(const zd_params *p, const zd_pk *pk, int64_t n, const int32_t *kxyz, double *out) {
    zd_plan *pl = nullptr;
    if (make_test_gen(p, pk, &pl)) return 1;
    int *d_k = nullptr;
    double *d_o = nullptr;
    int rc = 1;
    do {
        if (hipMalloc((void **) &d_k, sizeof(int) * 3 * n) != hipSuccess) break;
        if (hipMalloc((void **) &d_o, sizeof(double) * 3 * n) != hipSuccess) break;
        if (hipMemcpy(d_k, kxyz, sizeof(int) * 3 * n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (zd::launch_test_modes_table(pl->g, n, d_k, d_o, 0)) break;
        if (hipMemcpy(out, d_o, sizeof(double) * 3 * n, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    hipFree(d_k);
    hipFree(d_o);
    zd_plan_destroy(pl);
    return rc;
}

int zd_test_generate_loopback(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, zd_slab_cb cb, void *user,
                              zd_stats *out) {
    if (p->ngpu < 2) return 1;
    return zd_generate_multi(p, pk, eig, eig_ppd, cb, user, out, 2);
}

int zd_test_v1_words(int64_t seed, int32_t nblocks, uint32_t *out) {
    zd::V1Stream *d_s = nullptr;
    uint32_t *d_o = nullptr;
    int rc = 1;
    do {
        if (nblocks <= 0) break;
        if (hipMalloc((void **) &d_s, sizeof(zd::V1Stream)) != hipSuccess) break;
        if (hipMalloc((void **) &d_o, sizeof(uint32_t) * 624 * (size_t) nblocks) != hipSuccess) break;
        if (zd::launch_v1_seed((unsigned long long) seed, 1, d_s, 0)) break;
        if (zd::launch_test_v1_words(d_s, nblocks, d_o, 0)) break;
        if (hipMemcpy(out, d_o, sizeof(uint32_t) * 624 * (size_t) nblocks, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    hipFree(d_s);
    hipFree(d_o);
    return rc;
}

static int test_fft_composite(int32_t n, int64_t lines, int32_t axis_kind, const double *in, double *out) {
    int P = 0, Q = 0;
    const int W = zd::test_fftq_tile_width(n);
    if (!zd::np2_split(n, &P, &Q) || W == 0 || lines % W) {
        fprintf(stderr, "zd_test_fft: n=%d is not 2^a * {3, 9, 27, 5, 15, 25, 45, 75, 125, 135, 7, 21, 35, 49}, or lines %% %d != 0\n", n, W);
        return 1;
    }
    std::vector<cplx> twP = make_twiddles(P), twN = make_twiddles(n), twQ = make_twiddles(Q);
    cplx *d_tw = nullptr, *d_in = nullptr, *d_out = nullptr;
    const size_t nb = sizeof(cplx) * (size_t) n * lines;
    int rc = 1;
    do {
        if (hipMalloc((void **) &d_tw, sizeof(cplx) * (P + n + Q)) != hipSuccess) break;
        if (hipMalloc((void **) &d_in, nb) != hipSuccess) break;
        if (hipMalloc((void **) &d_out, nb) != hipSuccess) break;
        if (hipMemcpy(d_tw, twP.data(), sizeof(cplx) * P, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_tw + P, twN.data(), sizeof(cplx) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_tw + P + n, twQ.data(), sizeof(cplx) * Q, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_in, in, nb, hipMemcpyHostToDevice) != hipSuccess) break;
        if (zd::launch_test_fftq(n, axis_kind, d_tw, d_tw + P, d_tw + P + n, d_in, d_out, lines, 0)) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(out, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    hipFree(d_tw);
    hipFree(d_in);
    hipFree(d_out);
    return rc;
}

// any length through the convolution kernels (zd_kernels_any.hip); axis_kind as in zd_test_fft
static int test_fft_any(int32_t n, int64_t lines, int32_t axis_kind, const double *in, double *out) {
    zd::AnyTab tab = {};
    cplx *bufs[3] = {nullptr, nullptr, nullptr}, *d = nullptr;
    const size_t nb = sizeof(cplx) * (size_t) n * lines;
    int rc = 1;
    do {
        if (n < 2 || n > 8192) break;
        if (any_make_tab(n, &tab, bufs)) break;
        if (hipMalloc((void **) &d, nb) != hipSuccess) break;
        if (hipMemcpy(d, in, nb, hipMemcpyHostToDevice) != hipSuccess) break;
        if (axis_kind == 1 ? zd::launch_any_cols(tab, d, 0, lines, (int) lines, 1, -1, 0) : zd::launch_any_lines(tab, d, n, lines, 0)) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(out, d, nb, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    for (cplx *b : bufs) hipFree(b);
    hipFree(d);
    return rc;
}

int zd_test_fft(int32_t n, int64_t lines, int32_t axis_kind, const double *in, double *out) {
    if (!is_pow2(n)) {
        int P = 0, Q = 0;
        // composite-length kernels where they exist and the batch is whole tiles; else the convolution kernels (any length,
        // ragged batches)
        if (zd::np2_split(n, &P, &Q) && zd::test_fftq_tile_width(n) > 0 && P >= 8 && lines % zd::test_fftq_tile_width(n) == 0)
            return test_fft_composite(n, lines, axis_kind, in, out);
        return test_fft_any(n, lines, axis_kind, in, out);
    }
    const int W = zd::test_fft_tile_width(n);
    if (W == 0 || lines % W) {
        fprintf(stderr, "zd_test_fft: n=%d needs lines %% %d == 0\n", n, W);
        return 1;
    }
    std::vector<cplx> tw = make_twiddles(n);
    cplx *d_tw = nullptr, *d_in = nullptr, *d_out = nullptr;
    const size_t nb = sizeof(cplx) * (size_t) n * lines;
    int rc = 1;
    do {
        if (hipMalloc((void **) &d_tw, sizeof(cplx) * n) != hipSuccess) break;
        if (hipMalloc((void **) &d_in, nb) != hipSuccess) break;
        if (hipMalloc((void **) &d_out, nb) != hipSuccess) break;
        if (hipMemcpy(d_tw, tw.data(), sizeof(cplx) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_in, in, nb, hipMemcpyHostToDevice) != hipSuccess) break;
        if (zd::launch_test_fft(n, axis_kind, d_tw, d_in, d_out, lines, 0)) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(out, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (0);
    hipFree(d_tw);
    hipFree(d_in);
    hipFree(d_out);
    return rc;
}

#endif  // ZD_TESTING

#ifdef ZD_TUNING
// tuning harness (not part of the product path): time y-pass tile variants on a synthetic store
int zd_test_yfft_variant(int32_t n, int32_t variant, int32_t narray, int32_t nplanes, int32_t tiled, int32_t reps,
                         double *ms_per_launch) {
    zd::StoreLayout S;
    S.N = n; S.half = n / 2; S.Hq = n / 2; S.narray = narray;
    S.lHq = 0;
    while ((1 << S.lHq) < S.Hq) S.lHq++;
    S.lG = 0;
    S.ky_stride = 1;
    int lBk = 0, lBz = 0;
    if (tiled) {
        const long long target = std::max<long long>(1, ((long long) 2 << 20) / ((long long) n * 16));
        int lt = 0;
        while ((1LL << (lt + 1)) <= target) lt++;
        lBk = (lt + 1) / 2; lBz = lt - lBk;
        while ((1 << lBz) > nplanes) lBz--;
    }
    S.lq = 0;
    S.lBk = lBk; S.lBz = lBz; S.rows_outer = 0; S.one_block = 0; S.pitch = n; S.prune = 0; S.nt = 0; S.kmax = 0; S.fund2 = 0; S.k2_cutoff = 0;
    S.a_rows = (1 << lBk) << lBz;
    S.zb_rows = S.a_rows * narray;
    S.kb_rows = S.zb_rows * (nplanes >> lBz);
    S.chunk_rows = S.kb_rows * (n >> lBk);
    std::vector<cplx> tw = make_twiddles(n);
    cplx *d_tw = nullptr, *d = nullptr;
    const size_t nb = (size_t) S.chunk_rows * S.pitch * 16;
    int rc = 1;
    do {
        if (hipMalloc((void **) &d_tw, sizeof(cplx) * n) != hipSuccess) break;
        if (hipMalloc((void **) &d, nb) != hipSuccess) break;
        hipMemcpy(d_tw, tw.data(), sizeof(cplx) * n, hipMemcpyHostToDevice);
        hipMemset(d, 0, nb);
        if (zd::launch_yfft_variant(variant, S, nplanes, d_tw, d, 0)) break;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; i++) zd::launch_yfft_variant(variant, S, nplanes, d_tw, d, 0);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *ms_per_launch = ms / reps;
        hipEventDestroy(e0); hipEventDestroy(e1);
        rc = 0;
    } while (0);
    hipFree(d_tw); hipFree(d);
    return rc;
}

int zd_test_copy_bw(int64_t bytes, int32_t reps, double *gbps) {
    void *a = nullptr, *b = nullptr;
    int rc = 1;
    do {
        if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) break;
        hipMemset(a, 1, bytes);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        zd::launch_copy16(a, b, bytes / 16, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; i++) zd::launch_copy16(a, b, bytes / 16, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *gbps = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        rc = 0;
    } while (0);
    hipFree(a);
    hipFree(b);
    return rc;
}

#endif  // ZD_TUNING

}  // extern "C"
