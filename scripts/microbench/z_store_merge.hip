// Go / no-go (i) of a generator fused with the z FFT, lanes along k2: the STORE pattern such a kernel would have.
// One wave owns one (row, column) line of L = 512 planes, lane t holds z2 = t + 64 e (e < 8); a workgroup = the 8 rows of a row
// block (8 waves), 4 fields per column.  Field-store address of (plane z2, field f, row block b, column x, row r):
//     ((z2 * 4 + f) * field_elems + (b * cols + x) * 8 + r) * 16 B       — 8 rows x 16 B = one 128-byte line per (z2, f, b, x)
//  direct : every wave stores its own 16-byte pieces (one instruction = 64 pieces in 64 different planes; the 8 waves of the
//           workgroup complete a line between them, if the L2 merges them)
//  merged : the 8 waves stage 64 planes x 8 rows x 16 B in LDS, then every store instruction writes 8 whole 128-byte lines
//  zfft_f : what k_zfft_f<512,16,1> does today (256 threads: 8 rows x 32 t-values, z2 = t + 32 e): 8 whole lines per instruction
//   hipcc --offload-arch=gfx950 -O3 z_store_merge.hip -o z_store_merge && ./z_store_merge
#include <hip/hip_runtime.h>
#include <stdio.h>
struct cplx { double x, y; };
constexpr int L = 512, NF = 4;

template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 256 : 512) void k_store(cplx *__restrict__ out, long long field_elems, int cols) {
    __shared__ cplx stage[2][64 * 8];
    const long long lineidx = blockIdx.x;  // (row block, column)
    if constexpr (MODE == 2) {
        const int r = threadIdx.x & 7, t = threadIdx.x >> 3;  // 32 t-values, 16 elements each
        for (int f = 0; f < NF; f++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int z2 = t + 32 * e;
                out[((long long) z2 * NF + f) * field_elems + lineidx * 8 + r] = cplx{1.0 + e, 2.0 + t + f};
            }
        }
    } else {
        const int r = threadIdx.x >> 6, t = threadIdx.x & 63;  // wave = row
        for (int f = 0; f < NF; f++) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int z2 = t + 64 * e;
                const cplx v = cplx{1.0 + e, 2.0 + t + f};
                if constexpr (MODE == 0) {
                    out[((long long) z2 * NF + f) * field_elems + lineidx * 8 + r] = v;
                } else {
                    cplx *s = stage[e & 1];
                    s[t * 8 + r] = v;  // [plane of the group][row]
                    __syncthreads();
                    const int zz = threadIdx.x >> 3, rr = threadIdx.x & 7;  // 8 consecutive lanes = one line
                    out[((long long) (zz + 64 * e) * NF + f) * field_elems + lineidx * 8 + rr] = s[zz * 8 + rr];
                    // (double buffer: the next group's writes go to the other half; one barrier per group)
                }
            }
        }
    }
}

int main() {
    const int cols = 4096, blocks = 48;  // 48 row blocks x 4096 columns per (plane, field) image
    const long long field_elems = (long long) blocks * cols * 8;
    const size_t bytes = (size_t) L * NF * field_elems * 16;
    cplx *out;
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc of %.1f GB failed\n", bytes / 1e9); return 1; }
    hipMemset(out, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 3; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %6.0f GB/s  (%.1f ms for %.1f GB)\n", name, 3.0 * bytes / (ms * 1e-3) / 1e9, ms / 3, bytes / 1e9);
    };
    const unsigned grid = (unsigned) (blocks * cols);
    time("direct: 8 waves x 16-byte pieces of a line (L2 to merge)", [&] { k_store<0><<<grid, 512>>>(out, field_elems, cols); });
    time("merged through LDS: whole 128-byte lines, 8 per instruction", [&] { k_store<1><<<grid, 512>>>(out, field_elems, cols); });
    time("k_zfft_f<512,16,1> pattern (256 threads, 8 rows x 32 t)", [&] { k_store<2><<<grid, 256>>>(out, field_elems, cols); });
    return 0;
}
