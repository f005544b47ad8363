// TEST INFRASTRUCTURE ONLY (oracle/_ref): thin extern "C" driver around the REFERENCE's own
// header-only sources, compiled from where they lie under /root/reference:
//   include/pcg-rng/pcg_random.hpp   (pcg64 engine, advance, distance)
//   include/spline_function.h        (SplineFunction::load/spline/val)  [needs <fmt/base.h>: the
//                                     image's real fmt, bundled under torch/include, header-only]
// Nothing here restates reference arithmetic; it only calls it.  Built into oracle/_ref/ by
// oracle/Makefile (target `ref`) when /root/reference is present.  Used by
// tests/golden/make_golden.py to generate committed known-answer vectors and by tests to pin
// oracle/zd_oracle.c.  The rest of the reference (zeldovich.cpp, power_spectrum.cpp, output.cpp,
// block_array.cpp, parameters.cpp) needs FFTW3, GSL and flex/bison-generated ParseHeader code that
// this image lacks, so it is unbuildable here and is NOT part of this build.
#include <cstdint>
#include <cstring>
#include "pcg-rng/pcg_random.hpp"
#include "spline_function.h"

// state is returned/accepted as two u64 (hi, lo).  state_ is a protected member of the reference
// engine; a derived struct exposes it without touching the reference header.
struct pcg64_open : public pcg64 {
    using pcg64::pcg64;
    pcg64_open() : pcg64() {}
    pcg64_open(const pcg64 &g) : pcg64(g) {}
    __uint128_t get() const { return state_; }
    void set(__uint128_t s) { state_ = s; }
};
static pcg64 from_state(uint64_t hi, uint64_t lo) {
    pcg64_open g;
    g.set((((__uint128_t) hi) << 64) | lo);
    return g;
}
static void get_state(const pcg64 &g, uint64_t *hi, uint64_t *lo) {
    pcg64_open o(g);
    __uint128_t s = o.get();
    *hi = (uint64_t)(s >> 64);
    *lo = (uint64_t) s;
}

extern "C" {

void ref_pcg_seed(uint64_t seed, uint64_t *hi, uint64_t *lo) {
    pcg64 g(seed);
    get_state(g, hi, lo);
}
// draws n values starting from state (hi,lo); returns new state
void ref_pcg_draw(uint64_t *hi, uint64_t *lo, int n, uint64_t *out) {
    pcg64 g = from_state(*hi, *lo);
    for (int i = 0; i < n; i++) out[i] = g();
    get_state(g, hi, lo);
}
void ref_pcg_advance(uint64_t *hi, uint64_t *lo, uint64_t delta_hi, uint64_t delta_lo) {
    pcg64 g = from_state(*hi, *lo);
    __uint128_t d = (((__uint128_t) delta_hi) << 64) | delta_lo;
    g.advance(d);
    get_state(g, hi, lo);
}
// distance b - a (as pcg's operator-), low 64 bits
uint64_t ref_pcg_distance(uint64_t ahi, uint64_t alo, uint64_t bhi, uint64_t blo) {
    pcg64 a = from_state(ahi, alo), b = from_state(bhi, blo);
    return (uint64_t)(b - a);
}

// Natural cubic spline through n nodes (x,y) as the reference builds it; returns y2 and sorted x,y
void ref_spline_build(int n, const double *x, const double *y, double *xs, double *ys, double *y2s) {
    SplineFunction s(n + 8);
    for (int i = 0; i < n; i++) s.load(x[i], y[i]);
    s.spline();
    for (int i = 0; i < n; i++) s.get_node(i, &xs[i], &ys[i]);
    // y2 is private: recover it exactly is not possible through the API; instead expose val()
    (void) y2s;
}
void ref_spline_val(int n, const double *x, const double *y, int m, const double *v, double *out) {
    SplineFunction s(n + 8);
    for (int i = 0; i < n; i++) s.load(x[i], y[i]);
    s.spline();
    for (int i = 0; i < m; i++) out[i] = s.val(v[i]);
}
}
