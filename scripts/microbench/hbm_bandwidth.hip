#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d2v __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ void k_copy(const d2v* __restrict__ in, d2v* __restrict__ out, long long n) {
    long long i = ((long long) blockIdx.x * blockDim.x * U) + threadIdx.x;
    const long long s = (long long) gridDim.x * blockDim.x * U;
    for (; i + (U - 1) * blockDim.x < n; i += s) {
        d2v v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + u * blockDim.x) : in[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * blockDim.x); else out[i + u * blockDim.x] = v[u]; }
    }
}
template <int U>
__global__ void k_read(const d2v* __restrict__ in, d2v* __restrict__ out, long long n) {
    long long i = ((long long) blockIdx.x * blockDim.x * U) + threadIdx.x;
    const long long s = (long long) gridDim.x * blockDim.x * U;
    d2v acc = {0, 0};
    for (; i + (U - 1) * blockDim.x < n; i += s) {
#pragma unroll
        for (int u = 0; u < U; u++) acc += in[i + u * blockDim.x];
    }
    if (acc.x == 1.2345) out[0] = acc;
}
template <int U>
__global__ void k_write(d2v* __restrict__ out, long long n) {
    long long i = ((long long) blockIdx.x * blockDim.x * U) + threadIdx.x;
    const long long s = (long long) gridDim.x * blockDim.x * U;
    d2v v = {1.0, 2.0};
    for (; i + (U - 1) * blockDim.x < n; i += s) {
#pragma unroll
        for (int u = 0; u < U; u++) out[i + u * blockDim.x] = v;
    }
}
int main() {
    const long long bytes = 8LL << 30, n = bytes / 16;
    d2v *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch, double factor) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("BW %-28s %.0f GB/s\n", name, factor * bytes * 5 / (ms * 1e-3) / 1e9);
    };
    for (int g : {2048, 8192, 65536}) {
        printf("grid %d\n", g);
        time("copy U1", [&] { k_copy<1, false><<<g, 256>>>(a, b, n); }, 2);
        time("copy U4", [&] { k_copy<4, false><<<g, 256>>>(a, b, n); }, 2);
        time("copy U8", [&] { k_copy<8, false><<<g, 256>>>(a, b, n); }, 2);
        time("copy U4 nt", [&] { k_copy<4, true><<<g, 256>>>(a, b, n); }, 2);
        time("read U4", [&] { k_read<4><<<g, 256>>>(a, b, n); }, 1);
        time("write U4", [&] { k_write<4><<<g, 256>>>(b, n); }, 1);
    }
    return 0;
}
