#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// throughput of a few VALU instructions on gfx950: 8 independent chains per thread, 1 wave/SIMD vs 4 waves/SIMD
template <int OP>
__global__ void k(uint64_t* out, int iters, double seed) {
    uint64_t a[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 977 + i * 131 + 12345; d[i] = seed + i * 0.001 + threadIdx.x * 1e-6; }
    uint32_t m = 0x9E3779B1u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"((uint32_t) a[i]), "v"(m) : "vcc");
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(seed));
            if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(((uint32_t*) &a[i])[0]) : "v"(m));
            if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            if (OP == 4) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(((uint32_t*) &a[i])[0]));
            if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[i]));
            if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(((uint32_t*) &a[i])[0]) : "v"(m) : "vcc");
            if (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
            if (OP == 8) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[i]));
            if (OP == 9) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
        }
    }
    uint64_t s = 0; double t = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; t += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (uint64_t) t;
}
template <int OP> void run(const char* name, uint64_t* out) {
    const int iters = 20000;
    for (int wpb : {64, 256, 512}) {   // 1, 4, 8 waves per CU-block; one block per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<256, wpb>>>(out, 100, 1.0000001); hipDeviceSynchronize();
        hipEventRecord(e0); k<OP><<<256, wpb>>>(out, iters, 1.0000001); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double) iters * 8 * (wpb / 64.0) / 4.0;  // waves spread over 4 SIMDs
        printf("UB %-16s threads/block %4d: %.2f ns per wave-instr per SIMD (= %.1f cycles at 2.4 GHz)\n", name, wpb,
               ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    }
}
int main() {
    uint64_t* out; hipMalloc(&out, 256 * 512 * 8);
    run<0>("v_mad_u64_u32", out); run<1>("v_fma_f64", out); run<2>("v_mul_lo_u32", out); run<3>("v_rcp_f64", out);
    run<4>("v_cvt_f64_u32", out); run<5>("v_ldexp_f64", out); run<6>("v_add_co_u32", out); run<7>("v_mul_f64", out);
    run<8>("v_frexp_mant_f64", out); run<9>("v_lshlrev_b64", out);
    return 0;
}
