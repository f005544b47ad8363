// Go / no-go (round 5, VERDICT r4 #1b): the STORE side of a generator fused with the z FFT for the packed PLT3 store at PPD=2048
// (BASELINE C3: L = 1024 planes per pass, 3 arrays, rows of pitch N + 24, self row ky and twin row N - ky).
// A fused workgroup (512 threads, one per CU: its z lines fill the register file) walks a row along kx; one wave owns one
// (job, column) line, lane t holds planes t + 64 e.  After a DPP merge of two neighbouring columns one store instruction writes
// 32 runs of W x 16 bytes, one per plane.  What this measures:
//   * W = 1, 2, 4, 8 columns per run (W = 2 is what fits a CU: 2 x 6 lines x 16 KB = the whole register file's worth of lines)
//   * twin runs (row N - ky, columns N - x - W + 1 .. N - x) aligned to W (paired across tile boundaries) or off by one column
//   * the same stores behind `nfma` dependent FMAs per column (the generator's vector work): do the stores hide behind it?
//   hipcc -w --offload-arch=gfx950 -O3 fused_plt_store.hip -o fused_plt_store && ./fused_plt_store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct cplx { double x, y; };
constexpr int N = 2048, L = 1024, NA = 3, PITCH = N + 24, SEG = 256;

// element (plane, array, row, x) of the block store of one pass, single rank: [plane][array][row][x]
__device__ __forceinline__ long long at(int plane, int a, int row, int x) { return (((long long) plane * NA + a) * N + row) * PITCH + x; }

template <int W, bool TWIN_ALIGNED>
__global__ __launch_bounds__(512) void k_fused_store(cplx *__restrict__ out, int row0, int nrows, int nfma, int do_store, unsigned *ctr,
                                                     double *sink) {
    extern __shared__ double lds[];  // 118 KB requested: one workgroup per CU like the fused kernel
    __shared__ unsigned slot;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned nitems = (unsigned) (nrows * (N / SEG));
    double acc = threadIdx.x * 1e-9;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) slot = atomicAdd(ctr, 1u);
        __syncthreads();
        const unsigned item = slot;
        if (item >= nitems) break;
        const int ky = row0 + (int) (item / (N / SEG)), x0 = (int) (item % (N / SEG)) * SEG;
        // a "step" = W columns generated, transformed and stored
        for (int x = x0; x < x0 + SEG; x += W) {
            for (int i = 0; i < nfma * W; i++) acc = fma(acc, 1.0000001, 1e-9);  // the generator + transform of W columns
            __syncthreads();
            if (wave < 6 && do_store) {
                const int a = wave >> 1, twin = wave & 1;
                const int sub = lane % W, grp = lane / W;  // W lanes write one run
                constexpr int PPI = 64 / W;                 // planes per instruction
                int row, col;
                if (!twin) {
                    row = ky;
                    col = x + sub;
                } else {
                    row = N - ky;
                    const int xs = TWIN_ALIGNED ? x : x + 1;  // off by one: the run straddles the W-column boundary
                    col = (N - xs - W + sub + N) & (N - 1);
                    if (!TWIN_ALIGNED && W == 1) col = (N - x) & (N - 1);
                }
                const cplx v = cplx{acc + lane, acc - wave};
#pragma unroll 4
                for (int k = 0; k < L / PPI; k++) {  // the line's 1024 planes, PPI per instruction, for each of the W columns
                    const int plane = grp + PPI * k;
                    out[at(plane, a, row, col)] = v;
                }
            }
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

template <int W, bool TA>
static void run(const char *name, cplx *out, int rows, int nfma, int do_store, unsigned *ctr, double *sink) {
    hipFuncSetAttribute((const void *) k_fused_store<W, TA>, hipFuncAttributeMaxDynamicSharedMemorySize, 118 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(ctr, 0, 4);
        hipEventRecord(e0);
        k_fused_store<W, TA><<<256, 512, 118 * 1024>>>(out, 1, rows, nfma, do_store, ctr, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double) rows * N * L * 6 * 16;
    const double us_per_col = best * 1e3 / ((double) rows * N / 256.0);
    printf("%-58s W=%d nfma=%5d store=%d  %8.2f ms  %6.0f GB/s  %6.2f us per column and CU\n", name, W, nfma, do_store, best,
           do_store ? bytes / (best * 1e-3) / 1e9 : 0.0, us_per_col);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 96;
    const size_t bytes = (size_t) L * NA * N * PITCH * 16;
    cplx *out;
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc of %.1f GB failed\n", bytes / 1e9); return 1; }
    unsigned *ctr;
    double *sink;
    hipMalloc(&ctr, 4);
    hipMalloc(&sink, 8);
    printf("store %.1f GB, %d rows per run (%.1f GB written per run)\n", bytes / 1e9, rows, (double) rows * N * L * 96 / 1e9);
    run<1, true>("16-byte pieces", out, rows, 0, 1, ctr, sink);
    run<2, true>("32-byte runs, twin runs aligned", out, rows, 0, 1, ctr, sink);
    run<2, false>("32-byte runs, twin runs off by one column", out, rows, 0, 1, ctr, sink);
    run<4, true>("64-byte runs, aligned", out, rows, 0, 1, ctr, sink);
    run<4, false>("64-byte runs, twin off by one", out, rows, 0, 1, ctr, sink);
    run<8, true>("128-byte runs, aligned", out, rows, 0, 1, ctr, sink);
    // stores behind vector work: nfma dependent FMAs per column and thread (2 waves per SIMD: ~8 cycles per FMA and wave pair)
    for (int nfma : {1000, 2000, 3000, 4000}) {
        run<2, true>("compute only", out, rows, nfma, 0, ctr, sink);
        run<2, true>("compute + 32-byte runs (aligned twins)", out, rows, nfma, 1, ctr, sink);
    }
    run<2, false>("compute + 32-byte runs (twins off by one)", out, rows, 3000, 1, ctr, sink);
    return 0;
}
