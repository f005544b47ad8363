// Write bandwidth of the x stage's record pattern: a lane owns one 32-byte record and writes it as two 16-byte stores
// (records of consecutive lanes are adjacent: every store instruction of a wave covers 2 KB, half of each 32 bytes), against the
// same bytes written 16 contiguous bytes per lane (1 KB per instruction, whole).  Pure stores, 8 GB.
//   hipcc --offload-arch=gfx950 -O3 record_store.hip -o record_store && ./record_store
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(768) void k_rec(uint4 *__restrict__ out, long long nrec) {
    // a workgroup writes 4096 x 2 records like k_xfft<4096,16,3,1> (lines 0 and 1: 256 threads x 16 records each)
    const int t = threadIdx.x % 256, line = threadIdx.x / 256;
    if (line == 2) return;
    uint4 *base = out + ((long long) blockIdx.x * 2 + line) * 4096 * 2;
    uint4 q0 = {1u, 2u, (unsigned) t, 4u}, q1 = {5u, 6u, 7u, (unsigned) blockIdx.x};
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int xx = t + 256 * e;
        if (MODE == 0) {  // one lane per record: [2 xx], [2 xx + 1]
            base[2 * xx]     = q0;
            base[2 * xx + 1] = q1;
        } else {  // 16 contiguous bytes per lane per instruction: wave-instruction A = first KB of the wave's 2 KB, B = second
            const int w0 = (xx & ~63) * 2, l = xx & 63;
            base[w0 + l]      = q0;
            base[w0 + 64 + l] = q1;
        }
        q0.x += 1;
    }
}
int main() {
    const long long rows = 1 << 15;  // x 2 lines x 4096 records x 32 B = 8.6 GB
    const long long nrec = rows * 2 * 4096;
    uint4 *b;
    hipMalloc(&b, nrec * 32);
    hipMemset(b, 0, nrec * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-56s %6.0f GB/s\n", name, (double) nrec * 32 * 5 / (ms * 1e-3) / 1e9);
    };
    time("records, one lane per record (2 x 16 B at 32 B stride)", [&] { k_rec<0><<<rows, 768>>>(b, nrec); });
    time("records, 16 contiguous bytes per lane and instruction", [&] { k_rec<1><<<rows, 768>>>(b, nrec); });
    time("records, one lane per record (2 x 16 B at 32 B stride)", [&] { k_rec<0><<<rows, 768>>>(b, nrec); });
    return 0;
}
