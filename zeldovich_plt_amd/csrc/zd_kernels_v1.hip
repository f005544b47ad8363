// zd_kernels_v1.hip — ZD_Version = 1 random streams (legacy phases; src/power_spectrum.cpp:18-25,276-280,310-332 and the
// `ver == 1` branch of LoadPlane, src/zeldovich.cpp:365-370).
//
// Version 1 keeps ONE gsl_rng_mt19937 per yres (seed + yres, block = PPD / NumBlock of them).  Stream yres serves the rows
// ky = yres, yres + block, yres + 2 block, ... in that order (ZeldovichZ walks yblock outermost, zeldovich.cpp:558-571); inside
// a row the modes are visited z-major / x-minor and only the modes that survive the zero rule call cgauss<1>, which draws
// PAIRS of uniforms until one lands inside the unit circle.  So the n-th live mode of a stream owns the n-th accepted pair:
// there is no counter to jump to (the reason version 2 exists).  What is parallel:
//   * the block/G streams of the rows generated together (one workgroup per row);
//   * inside a stream, the MT19937 recurrence (three dependent sections of <= 227 words per 624-word regeneration) and
//     the accept / compact / hand-out steps, done with workgroup-wide prefix counts.
// k_v1_draw leaves the accepted (phase1, phase2) pairs in a dense [row][z][x] buffer; the general generator kernel
// (k_gen, zd_kernels.hip) turns them into D(k) = (phase1, phase2) * sqrt(-P ln r2 / r2) and does everything else as for
// version 2.  GSL is a system package of the reference, absent here: the generator is MT19937 (Matsumoto & Nishimura) with
// the 2002 initialisation exactly as gsl rng/mt.c has it — seed 0 -> 4357, gsl_rng_uniform = word / 2^32.
#include <hip/hip_runtime.h>
#include "zd_launch.h"

namespace zd {

constexpr int V1_T = 256;  // threads per stream workgroup (>= 227: one MT section per step)

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t k) {
    k ^= k >> 11;
    k ^= (k << 7) & 0x9d2c5680u;
    k ^= (k << 15) & 0xefc60000u;
    k ^= k >> 18;
    return k;
}
// The 624-word regeneration of mt_get (gsl rng/mt.c) on the LDS copy of the state.  Word i needs the OLD words i, i + 1
// and word i + 397 (mod 624), which is old for i < 227 and new otherwise: three sections, each read -> barrier -> write.
// Called by every thread of the workgroup.
__device__ __forceinline__ void mt_regenerate(uint32_t *mt) {
    const int t = threadIdx.x;
    uint32_t v  = 0;
    if (t < 227) v = mt_twist(mt[t], mt[t + 1], mt[t + 397]);
    __syncthreads();
    if (t < 227) mt[t] = v;
    __syncthreads();
    if (t < 227) v = mt_twist(mt[t + 227], mt[t + 228], mt[t]);
    __syncthreads();
    if (t < 227) mt[t + 227] = v;
    __syncthreads();
    if (t < 170) v = mt_twist(mt[t + 454], mt[t + 454 == 623 ? 0 : t + 455], mt[t + 227]);
    __syncthreads();
    if (t < 170) mt[t + 454] = v;
    __syncthreads();
}

// exclusive prefix of per-thread counts given the wave-local prefix / total: `base` = items of the lower waves
__device__ __forceinline__ void block_prefix(int wave_total, int *wsum, int &base, int &total) {
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) wsum[wave] = wave_total;
    __syncthreads();
    base = total = 0;
#pragma unroll
    for (int w = 0; w < V1_T / 64; w++) {
        const int c = wsum[w];
        base += w < wave ? c : 0;
        total += c;
    }
    __syncthreads();
}

// gsl_rng_set (mt_set): one thread per stream
__global__ void k_v1_seed(unsigned long long seed, int block, V1Stream *streams) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= block) return;
    unsigned long long sd = seed + (unsigned long long) s;  // longseed + i, power_spectrum.cpp:23
    if (sd == 0) sd = 4357;
    uint32_t w = (uint32_t) (sd & 0xffffffffULL);
    streams[s].mt[0] = w;
    for (int i = 1; i < 624; i++) {
        w = 1812433253u * (w ^ (w >> 30)) + (uint32_t) i;
        streams[s].mt[i] = w;
    }
    streams[s].nq = 0;
}

// One workgroup per row ky = ky0 + blockIdx.x * ky_stride; the rows of a launch belong to different streams.
//   dev[(blockIdx.x * N + z) * N + x] = (phase1, phase2) of the live modes of the row (other positions are not written)
__global__ __launch_bounds__(V1_T) void k_v1_draw(GenConst g, int block, int ky0, int ky_stride, V1Stream *__restrict__ streams,
                                                  double2 *__restrict__ dev, int *__restrict__ err) {
    __shared__ uint32_t mt[624];
    __shared__ double2 q[V1_QCAP];
    __shared__ int wsum[V1_T / 64];
    const int t = threadIdx.x, lane = t & 63;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const int N = g.N, half = g.half;
    const int ky = ky0 + (int) blockIdx.x * ky_stride;
    V1Stream &S = streams[ky % block];
    for (int i = t; i < 624; i += V1_T) mt[i] = S.mt[i];
    int count = (int) S.nq, head = 0;  // accepted pairs waiting in q[head .. head + count)  (workgroup-uniform)
    for (int i = t; i < count; i += V1_T) q[i] = S.q[i];
    __syncthreads();
    double2 *out = dev + (long long) blockIdx.x * N * N;
    const int ay = ky < 0 ? -ky : ky;
    for (int z = 0; z < N; z++) {
        const int kz = z > half ? z - N : z, az = kz < 0 ? -kz : kz;
        // lines without a live mode: the rule at kx = 0 (zeldovich.cpp:350-356) — uniform, no barrier inside
        if (ay == g.kmax || az == g.kmax) continue;
        if (!g.corner_modes && (double) (ky * ky + kz * kz) * g.fundamental2 >= g.k2_cutoff) continue;
        if (g.qonemode && (ky != g.one_mode[1] || kz != g.one_mode[2])) continue;
        for (int x0 = 0; x0 < N; x0 += V1_T) {
            const int x = x0 + t, kx = x > half ? x - N : x;
            const bool live = x < N && !mode_is_zero(g, kx, ky, kz, (double) (kx * kx + ky * ky + kz * kz) * g.fundamental2);
            const unsigned long long b = __ballot(live);
            int base, total;
            block_prefix(__popcll(b), wsum, base, total);
            if (total == 0) continue;
            for (int guard = 0; count < total; guard++) {  // refill: one regeneration = 312 attempts, ~245 accepted
                if (guard >= 64) {  // cannot happen with a working generator; leave instead of spinning
                    if (t == 0) *err = 1;
                    return;
                }
                mt_regenerate(mt);
                bool a0 = false, a1 = false;
                double2 v0 = {0, 0}, v1 = {0, 0};
                if (t < 156) {  // attempts 2t, 2t + 1 = words 4t .. 4t + 3; gsl_rng_uniform * 2 - 1 is exact
                    v0.x = mt_temper(mt[4 * t]) / 4294967296.0 * 2.0 - 1.0;
                    v0.y = mt_temper(mt[4 * t + 1]) / 4294967296.0 * 2.0 - 1.0;
                    v1.x = mt_temper(mt[4 * t + 2]) / 4294967296.0 * 2.0 - 1.0;
                    v1.y = mt_temper(mt[4 * t + 3]) / 4294967296.0 * 2.0 - 1.0;
                    const double r0 = __dadd_rn(__dmul_rn(v0.x, v0.x), __dmul_rn(v0.y, v0.y));
                    const double r1 = __dadd_rn(__dmul_rn(v1.x, v1.x), __dmul_rn(v1.y, v1.y));
                    a0 = r0 < 1.0 && r0 > 0.0;
                    a1 = r1 < 1.0 && r1 > 0.0;
                }
                const unsigned long long b0 = __ballot(a0), b1 = __ballot(a1);
                int abase, atotal;
                block_prefix(__popcll(b0) + __popcll(b1), wsum, abase, atotal);
                const int pos = head + count + abase + __popcll(b0 & lt) + __popcll(b1 & lt);
                if (a0) q[pos & (V1_QCAP - 1)] = v0;
                if (a1) q[(pos + (a0 ? 1 : 0)) & (V1_QCAP - 1)] = v1;
                count += atotal;
                __syncthreads();
            }
            if (live) out[(long long) z * N + x] = q[(head + base + __popcll(b & lt)) & (V1_QCAP - 1)];
            __syncthreads();  // the pairs are taken before a later refill reuses their slots
            head = (head + total) & (V1_QCAP - 1);
            count -= total;
        }
    }
    for (int i = t; i < 624; i += V1_T) S.mt[i] = mt[i];
    for (int i = t; i < count; i += V1_T) S.q[i] = q[(head + i) & (V1_QCAP - 1)];
    if (t == 0) S.nq = (uint32_t) count;
}

#ifdef ZD_TESTING
// test hook: the first 624 * nblocks tempered words of one stream (pins the parallel regeneration against the serial one)
__global__ __launch_bounds__(V1_T) void k_test_v1_words(V1Stream *streams, int nblocks, uint32_t *__restrict__ out) {
    __shared__ uint32_t mt[624];
    for (int i = threadIdx.x; i < 624; i += V1_T) mt[i] = streams[0].mt[i];
    __syncthreads();
    for (int b = 0; b < nblocks; b++) {
        mt_regenerate(mt);
        for (int i = threadIdx.x; i < 624; i += V1_T) out[(long long) b * 624 + i] = mt_temper(mt[i]);
        __syncthreads();
    }
}

#endif  // ZD_TESTING
int launch_v1_seed(unsigned long long seed, int block, V1Stream *streams, hipStream_t st) {
    hipLaunchKernelGGL(k_v1_seed, dim3((block + 63) / 64), dim3(64), 0, st, seed, block, streams);
    ZD_LAUNCH_CHECK();
    return 0;
}
int launch_v1_draw(const GenConst &g, int block, int ky0, int ky_stride, int nrows, V1Stream *streams, void *dev, int *err,
                   hipStream_t st) {
    hipLaunchKernelGGL(k_v1_draw, dim3(nrows), dim3(V1_T), 0, st, g, block, ky0, ky_stride, streams, (double2 *) dev, err);
    ZD_LAUNCH_CHECK();
    return 0;
}
#ifdef ZD_TESTING
int launch_test_v1_words(V1Stream *streams, int nblocks, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_test_v1_words, dim3(1), dim3(V1_T), 0, st, streams, nblocks, out);
    ZD_LAUNCH_CHECK();
    return 0;
}

#endif  // ZD_TESTING
}  // namespace zd
