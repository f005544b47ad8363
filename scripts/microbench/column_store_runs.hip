// Write bandwidth of the y stage's store pattern: a workgroup of W*N/16 threads owns W adjacent columns of an N x N plane
// (row pitch N + 24 complex doubles) and every thread stores its 16 rows (t + T e): runs of W x 16 bytes, one per row.
// Tile -> workgroup mapping as in k_yfft_f (the tiles of a 128-byte line consecutive on one XCD).  Also: the same bytes
// with a fixed number of reads per workgroup in front (a strided column read of a second plane), to see stores and loads mixed.
//   hipcc --offload-arch=gfx950 -O3 column_store_runs.hip -o column_store_runs && ./column_store_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
struct cplx { double x, y; };
template <int N, int W, bool READ, int E = 16>
__global__ __launch_bounds__(W *N / E) void k_cols(const cplx *__restrict__ in, cplx *__restrict__ out, int pitch) {
    constexpr int T = N / E, NT = N / W, TPL = W >= 8 ? 1 : 8 / W;
    const int w = threadIdx.x % W, t = threadIdx.x / W;
    const int id = blockIdx.x, xcd = id & 7, s = id >> 3;
    const int tile = ((s / TPL) * 8 + xcd) * TPL + s % TPL;  // the TPL tiles of a line: consecutive workgroups of one XCD
    const int x = tile * W + w;
    const size_t plane = (size_t) blockIdx.y * N * pitch;
    cplx v[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (READ) v[e] = in[plane + (size_t) (t + T * e) * pitch + x];
        else v[e] = cplx{1.0 + e, 2.0 + t};
    }
#pragma unroll
    for (int e = 0; e < E; e++) out[plane + (size_t) (t + T * e) * pitch + x] = cplx{v[e].x + 1.0, v[e].y};
    static_assert(NT % (8 * TPL) == 0, "tiles");
}
int main() {
    constexpr int N = 4096;
    const int pitch = N + 24, planes = 21;
    const size_t bytes = (size_t) planes * N * pitch * 16;
    cplx *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch, double factor) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 3; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %6.0f GB/s  (%.2f ms per plane)\n", name, factor * planes * (double) N * N * 16 * 3 / (ms * 1e-3) / 1e9, ms / 3 / planes);
    };
    // (8 and 16 columns per workgroup do not fit a CU with the transform's 16 elements per thread: 32 / 64 rows per thread here,
    // only to see what longer runs would buy)
    time("stores only, runs of 256 B (W=16, 64 rows/thr)", [&] { k_cols<N, 16, false, 64><<<dim3(N / 16, planes), 1024>>>(a, b, pitch); }, 1);
    time("stores only, runs of 128 B (W=8, 32 rows/thr)", [&] { k_cols<N, 8, false, 32><<<dim3(N / 8, planes), 1024>>>(a, b, pitch); }, 1);
    time("stores only, runs of 64 B  (W=4, 1024 thr)", [&] { k_cols<N, 4, false><<<dim3(N / 4, planes), 1024>>>(a, b, pitch); }, 1);
    time("stores only, runs of 32 B  (W=2,  512 thr)", [&] { k_cols<N, 2, false><<<dim3(N / 2, planes), 512>>>(a, b, pitch); }, 1);
    time("stores only, runs of 16 B  (W=1,  256 thr)", [&] { k_cols<N, 1, false><<<dim3(N / 1, planes), 256>>>(a, b, pitch); }, 1);
    time("column copy,  runs of 128 B (W=8, 32 rows/thr)", [&] { k_cols<N, 8, true, 32><<<dim3(N / 8, planes), 1024>>>(a, b, pitch); }, 2);
    time("column copy,  runs of 64 B  (W=4)", [&] { k_cols<N, 4, true><<<dim3(N / 4, planes), 1024>>>(a, b, pitch); }, 2);
    time("column copy,  runs of 32 B  (W=2)", [&] { k_cols<N, 2, true><<<dim3(N / 2, planes), 512>>>(a, b, pitch); }, 2);
    time("column copy,  runs of 16 B  (W=1)", [&] { k_cols<N, 1, true><<<dim3(N / 1, planes), 256>>>(a, b, pitch); }, 2);
    return 0;
}
