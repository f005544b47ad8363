// zd_pcg.h — pcg64 (setseq_xsl_rr_128_64) as a counter-addressable generator, host + device.
//
// Reference: include/pcg-rng/pcg_random.hpp (:159-170 multiplier/increment, :427-432 seeding,
// :381-386 + :855 advance-then-output for 128-bit state, :1144-1170 XSL-RR, :657-687 advance) and
// the per-plane stream layout of src/power_spectrum.cpp:26-37 + src/zeldovich.cpp:314-363:
// mode (kx, ky>=0, kz) consumes draws number c+1 and c+2 of the single stream, with
//     c = 2*((ky*65536 + (kz mod 65536))*65536 + (kx mod 65536)).
// On the GPU nothing is sequential: a lane jumps straight to its first mode with a table of
// 2^i-step affine maps and then walks with a fixed stride, which for an LCG is again ONE
// multiply-add (state' = A*state + C).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZD_HD __host__ __device__ __forceinline__
#else
#ifndef ZD_HD
#define ZD_HD inline __attribute__((always_inline))
#endif
#endif

namespace zdpcg {

typedef unsigned __int128 u128;

struct Affine {  // s -> A*s + C  (mod 2^128)
    u128 A, C;
};

#define ZD_PCG_MULT ((((zdpcg::u128) 0x2360ed051fc65da4ULL) << 64) | 0x4385df649fccf645ULL)
#define ZD_PCG_INC ((((zdpcg::u128) 0x5851f42d4c957f2dULL) << 64) | 0x14057b7ef767814fULL)

ZD_HD u128 step(u128 s) { return s * ZD_PCG_MULT + ZD_PCG_INC; }
ZD_HD u128 apply(const Affine &m, u128 s) { return m.A * s + m.C; }

ZD_HD uint64_t output(u128 s) {  // XSL-RR 128 -> 64
    uint64_t x   = (uint64_t) (s >> 64) ^ (uint64_t) s;
    unsigned rot = (unsigned) (s >> 122);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

// one_rand<2>: (0,1]  (src/power_spectrum.cpp:284-308)
ZD_HD double u01(uint64_t r) {
    if (r == 0xFFFFFFFFFFFFFFFFULL) return 1.0;
    r += 1ULL;
    return (double) r * 5.42101086242752217e-20;  // 2^-64, exact scaling (== ldexp(r,-64))
}

// counter of the first draw slot of mode (kx,ky,kz), ky >= 0
ZD_HD uint64_t mode_counter(int kx, int ky, int kz) {
    return 2ULL * ((((uint64_t) ky << 16) + (uint64_t) (kz & 65535)) * 65536ULL + (uint64_t) (kx & 65535));
}

// ---- host-side construction of jump maps ----
inline u128 seed_state(uint64_t seed) { return ((u128) seed + ZD_PCG_INC) * ZD_PCG_MULT + ZD_PCG_INC; }

// affine map for `delta` steps (delta taken mod 2^128; negative strides are 2^128 - |delta|)
inline Affine jump_map(u128 delta) {
    u128 cur_mult = ZD_PCG_MULT, cur_plus = ZD_PCG_INC, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    Affine m;
    m.A = acc_mult;
    m.C = acc_plus;
    return m;
}

constexpr int NBITS = 52;  // counters stay below 2^50 (ky < 2^15, 2*65536^2 per plane)
struct BitTable {
    Affine m[NBITS];  // m[i] advances by 2^i draws
};
inline void make_bit_table(BitTable &t) {
    for (int i = 0; i < NBITS; i++) t.m[i] = jump_map(((u128) 1) << i);
}

}  // namespace zdpcg
