// Does the 256 MiB Infinity Cache absorb a producer -> consumer hand-off between two kernels?  (The Z stage writes the folded
// inputs with the generator and reads them back with the z FFT: 1.75 TB per step at PPD=4096, 0.81 TB at PPD=2048 PLT.)
// A ring of two units of U bytes: W(i) writes unit i (16 B per lane, streaming); RW(i) reads unit i and writes U bytes of
// "store" elsewhere (never re-read).  serial: W(0) RW(0) W(1) RW(1) ... on one stream; overlapped: W(i+1) beside RW(i) on two
// streams (what the Z stage does with its slabs).  Rate = 3U per unit / time.  If small units run faster than large ones, the
// read-back is served on-die.
//   hipcc --offload-arch=gfx950 -O3 mall_handoff.hip -o mall_handoff && ./mall_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_w(uint4 *__restrict__ p, size_t n, unsigned tag) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        p[i] = uint4{tag, (unsigned) i, 3u, 4u};
}
__global__ void k_rw(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint4 v = in[i];
        v.x += 1u;
        out[i] = v;
    }
}
int main() {
    const size_t total = (size_t) 24 << 30;  // bytes of "store" written per measurement
    uint4 *ring, *store;
    hipMalloc(&ring, (size_t) 4 << 30);
    hipMalloc(&store, total);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    hipEvent_t e0, e1, evw[2], evr[2];
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) { hipEventCreateWithFlags(&evw[i], hipEventDisableTiming); hipEventCreateWithFlags(&evr[i], hipEventDisableTiming); }
    const int grid = 256 * 8;
    for (int mb : {8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048}) {
        const size_t U = (size_t) mb << 20, n = U / 16, units = total / U;
        for (int overlapped = 0; overlapped < 2; overlapped++) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s0);
            if (!overlapped) {
                for (size_t i = 0; i < units; i++) {
                    uint4 *u = ring + (i & 1) * n;
                    k_w<<<grid, 256, 0, s0>>>(u, n, (unsigned) i);
                    k_rw<<<grid, 256, 0, s0>>>(u, store + i * n, n);
                }
            } else {
                for (size_t i = 0; i < units; i++) {
                    uint4 *u = ring + (i & 1) * n;
                    if (i >= 2) hipStreamWaitEvent(s1, evr[i & 1], 0);  // the unit's previous content has been read
                    else if (i == 0) hipStreamWaitEvent(s1, e0, 0);
                    k_w<<<grid / 2, 256, 0, s1>>>(u, n, (unsigned) i);
                    hipEventRecord(evw[i & 1], s1);
                    hipStreamWaitEvent(s0, evw[i & 1], 0);
                    k_rw<<<grid / 2, 256, 0, s0>>>(u, store + i * n, n);
                    hipEventRecord(evr[i & 1], s0);
                }
            }
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("unit %5d MB  %-10s  %6.0f GB/s of write + read-back + store (3U per unit)   %.1f us per unit\n", mb,
                   overlapped ? "overlapped" : "serial", 3.0 * U * units / (ms * 1e-3) / 1e9, ms * 1e3 / units);
        }
    }
    return 0;
}
