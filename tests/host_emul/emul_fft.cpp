// Host emulation of the device FFT engine (zd_fft.h): runs the very same pass/exchange index
// arithmetic thread by thread on the CPU, phase by phase (a phase boundary stands for
// __syncthreads()).  Lets the CPU-only test suite validate the engine without a GPU.
#include <vector>
#include <cmath>
#include <cstring>
#include "../../zeldovich_plt_amd/csrc/zd_fft.h"
using namespace zdfft;

template <class PL, class LDS, int P>
static void run_passes(std::vector<double> &re, std::vector<double> &im, int W, const cplx *tw,
                       std::vector<double> &lds) {
    constexpr int E = PL::E, T = PL::T;
    const int nthreads = T * W;
    for (int tid = 0; tid < nthreads; tid++) {
        int w = tid % W, t = tid / W;
        pass_compute<PL, P>(*(double(*)[E]) & re[tid * E], *(double(*)[E]) & im[tid * E], t, tw);
        (void) w;
    }
    if constexpr (P + 1 < PL::NPASS) {
        for (int part = 0; part < 2; part++) {
            std::vector<double> &v = part == 0 ? re : im;
            for (int tid = 0; tid < nthreads; tid++) {
                int w = tid % W, t = tid / W;
                xchg_write<PL, P, LDS>(*(double(*)[E]) & v[tid * E], t, w, lds.data());
            }
            for (int tid = 0; tid < nthreads; tid++) {
                int w = tid % W, t = tid / W;
                xchg_read<PL, LDS>(*(double(*)[E]) & v[tid * E], t, w, lds.data());
            }
        }
        run_passes<PL, LDS, P + 1>(re, im, W, tw, lds);
    }
}

template <int N, int E, int W, bool LINE>
static void emul(const double *in, double *out) {
    using PL  = Plan<N, E>;
    using LDS = typename std::conditional<LINE, LineInner<N, W>, ColsInner<N, W>>::type;
    std::vector<cplx> tw(N);
    for (int k = 0; k < N; k++) {
        long double a = 2.0L * 3.14159265358979323846264338327950288L * k / N;
        tw[k].x = (double) cosl(a);
        tw[k].y = (double) sinl(a);
    }
    constexpr int T = PL::T;
    std::vector<double> re(T * W * E), im(T * W * E), lds(LDS::SIZE, 0.0);
    // in: [w][n] complex interleaved
    for (int tid = 0; tid < T * W; tid++) {
        int w = tid % W, t = tid / W;
        for (int e = 0; e < E; e++) {
            re[tid * E + e] = in[2 * (w * N + t + T * e)];
            im[tid * E + e] = in[2 * (w * N + t + T * e) + 1];
        }
    }
    run_passes<PL, LDS, 0>(re, im, W, tw.data(), lds);
    for (int tid = 0; tid < T * W; tid++) {
        int w = tid % W, t = tid / W;
        for (int e = 0; e < E; e++) {
            out[2 * (w * N + t + T * e)]     = re[tid * E + e];
            out[2 * (w * N + t + T * e) + 1] = im[tid * E + e];
        }
    }
}

extern "C" int emul_fft(int N, int E, int W, int line, const double *in, double *out) {
#define CASE(n, e, w)                                     \
    if (N == n && E == e && W == w) {                     \
        if (line) emul<n, e, w, true>(in, out);           \
        else emul<n, e, w, false>(in, out);               \
        return 0;                                         \
    }
    CASE(16, 16, 2) CASE(32, 16, 2) CASE(64, 16, 4) CASE(128, 16, 4) CASE(256, 16, 2) CASE(512, 16, 2)
    CASE(1024, 16, 2) CASE(2048, 16, 2) CASE(4096, 16, 1)
    CASE(8, 8, 2) CASE(64, 8, 4) CASE(128, 8, 2) CASE(256, 8, 2) CASE(512, 8, 2) CASE(2048, 8, 1)
    CASE(64, 4, 2) CASE(128, 4, 2) CASE(32, 2, 2) CASE(4, 4, 3) CASE(8, 4, 3) CASE(16, 4, 1)
    return 1;
}
