// zd_epi.h — device helpers of the particle epilogue shared by zd_kernels.hip and zd_kernels_np2.hip:
// one record of WriteParticlesSlab (src/output.cpp:128-141) and the workgroup reduction of max_disp / density_variance.
#pragma once
#include <hip/hip_runtime.h>

#include "zd_device.h"

namespace zd {}
using namespace zd;

__device__ __forceinline__ unsigned long long dbits(double v) { return (unsigned long long) __double_as_longlong(v); }

// one particle record (include/output.h:19-49; WriteParticlesSlab output.cpp:128-141: i = z, j = y, k = x,
// displ = (qz, qy, qx), vel = (vz, vy, vx)); pos/vel are in this code's x, y, z order
__device__ __forceinline__ void emit_record(char *__restrict__ records, long long pidx, const EpiConst &ec, int z, int yy,
                                            int xx, const double (&pos)[3], const double (&vel)[3], bool nt = false) {
    char *rec = records + pidx * ec.recsize;
    const unsigned int ij = ((unsigned int) z & 0xffffu) | (((unsigned int) yy & 0xffffu) << 16);
    const unsigned int k0 = ((unsigned int) xx & 0xffffu);
    if (ec.icformat == 1) {  // RVZel: u16 i,j,k + pad, float displ[3], vel[3]
        uint4 q0, q1;
        q0.x = ij;
        q0.y = k0;
        q0.z = __float_as_uint((float) pos[2]);
        q0.w = __float_as_uint((float) pos[1]);
        q1.x = __float_as_uint((float) pos[0]);
        q1.y = __float_as_uint((float) vel[2]);
        q1.z = __float_as_uint((float) vel[1]);
        q1.w = __float_as_uint((float) vel[0]);
        if (nt) {  // tuning knob (ZD_NT bit 7): streaming stores
            typedef unsigned int zd_u4v __attribute__((ext_vector_type(4)));
            zd_u4v a = {q0.x, q0.y, q0.z, q0.w}, b = {q1.x, q1.y, q1.z, q1.w};
            __builtin_nontemporal_store(a, reinterpret_cast<zd_u4v *>(rec));
            __builtin_nontemporal_store(b, reinterpret_cast<zd_u4v *>(rec) + 1);
        } else {
            reinterpret_cast<uint4 *>(rec)[0] = q0;
            reinterpret_cast<uint4 *>(rec)[1] = q1;
        }
    } else if (ec.icformat == 2) {  // RVdoubleZel: 56 B
        unsigned long long *r8 = reinterpret_cast<unsigned long long *>(rec);
        r8[0] = (unsigned long long) ij | ((unsigned long long) k0 << 32);
        double *d = reinterpret_cast<double *>(rec + 8);
        d[0] = pos[2];
        d[1] = pos[1];
        d[2] = pos[0];
        d[3] = vel[2];
        d[4] = vel[1];
        d[5] = vel[0];
    } else if (ec.icformat == 0) {  // Zeldovich: u16 i,j,k + pad, double displ[3]
        double2 q0, q1;
        q0.x = __longlong_as_double((long long) ((unsigned long long) ij | ((unsigned long long) k0 << 32)));
        q0.y = pos[2];
        q1.x = pos[1];
        q1.y = pos[0];
        reinterpret_cast<double2 *>(rec)[0] = q0;
        reinterpret_cast<double2 *>(rec)[1] = q1;
    } else {  // ZelSimple: float displ[3]
        float *d = reinterpret_cast<float *>(rec);
        d[0] = (float) pos[2];
        d[1] = (float) pos[1];
        d[2] = (float) pos[0];
    }
}


// max_disp of WriteParticlesSlab (src/output.cpp:28,190-193): per axis the SIGNED displacement of largest magnitude, and on a tie in
// magnitude the one met first in the reference's (z, y, x) loop order — `if (fabs(pos) > fabs(max)) max = pos` with a strict
// comparison.  Tracked as (signed value, linear lattice index (z N + y) N + x) per thread, reduced over the workgroup and combined
// globally by the key (|v|, smaller index).  (Rounds 1-3 kept max(+v) and max(-v) apart and let +v win a tie.)
struct MaxAbs {
    double v[3] = {0.0, 0.0, 0.0};
    unsigned long long lin[3] = {0ULL, 0ULL, 0ULL};
};
// strict form: the caller visits its records in increasing linear index (the first record of a magnitude stays)
__device__ __forceinline__ void max_track(MaxAbs &m, const double (&pos)[3], unsigned long long lin) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const bool g = fabs(pos[j]) > fabs(m.v[j]);
        m.v[j]   = g ? pos[j] : m.v[j];
        m.lin[j] = g ? lin : m.lin[j];
    }
}
// the same with a 32-bit tag local to the thread's row(s) (kernels held to 128 registers); widen() turns it into the lattice index
struct MaxAbs32 {
    double v[3] = {0.0, 0.0, 0.0};
    int tag[3] = {0, 0, 0};
};
__device__ __forceinline__ void max_track(MaxAbs32 &m, const double (&pos)[3], int tag) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const bool g = fabs(pos[j]) > fabs(m.v[j]);
        m.v[j]   = g ? pos[j] : m.v[j];
        m.tag[j] = g ? tag : m.tag[j];
    }
}
// any visiting order: ties in magnitude go to the smaller index
__device__ __forceinline__ bool max_better(double v, unsigned long long lin, double cur, unsigned long long curlin) {
    const double a = fabs(v), b = fabs(cur);
    return a > b || (a == b && a > 0.0 && lin < curlin);
}
__device__ __forceinline__ void max_track_any(MaxAbs32 &m, const double (&pos)[3], int tag) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const bool g = max_better(pos[j], (unsigned long long) (unsigned) tag, m.v[j], (unsigned long long) (unsigned) m.tag[j]);
        m.v[j]   = g ? pos[j] : m.v[j];
        m.tag[j] = g ? tag : m.tag[j];
    }
}
__device__ __forceinline__ void max_track_any(MaxAbs &m, const double (&pos)[3], unsigned long long lin) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const bool g = max_better(pos[j], lin, m.v[j], m.lin[j]);
        m.v[j]   = g ? pos[j] : m.v[j];
        m.lin[j] = g ? lin : m.lin[j];
    }
}

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one candidate of a workgroup into the replicated global slot: maxabs = bit pattern of |v| (monotone in |v| for v >= 0),
// maxkey = (linear index << 1) | (v < 0).  Almost every workgroup leaves after two loads (its |v| is below the slot's); a
// candidate that can improve the slot updates the pair under the slot's lock.  Call it from ONE lane of a wave at a time.
__device__ __forceinline__ void max_commit(Reduce *__restrict__ red, int j, int slot, double v, unsigned long long lin) {
    const unsigned long long myabs = dbits(fabs(v)), mykey = (lin << 1) | (v < 0.0 ? 1ULL : 0ULL);
    if (myabs == 0ULL) return;
    const unsigned long long cur = ld_agent(&red->maxabs[j][slot]);
    if (myabs < cur) return;
    // (pair read without the lock: the key of a given magnitude only decreases, and a larger magnitude beats this candidate anyway)
    if (myabs == cur && mykey >= ld_agent(&red->maxkey[j][slot])) return;
    // (bounded: a holder is a running wave a few memory operations from its release, so the bound is never reached; if it ever
    // were, an unlocked update — possibly torn against a concurrent one — is preferred to a wave that never finishes; such a
    // wave leaves the lock word alone, it belongs to the holder)
    bool held = false;
    for (unsigned spin = 0; spin < (1u << 22); spin++) {
        if (atomicCAS(&red->lock[slot], 0u, 1u) == 0u) {
            held = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    const unsigned long long a = ld_agent(&red->maxabs[j][slot]), k = ld_agent(&red->maxkey[j][slot]);
    if (myabs > a || (myabs == a && mykey < k)) {
        __hip_atomic_store(&red->maxkey[j][slot], mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&red->maxabs[j][slot], myabs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    if (held) atomicExch(&red->lock[slot], 0u);
}

// workgroup reduction of the epilogue -> one atomic per quantity into a replicated slot
template <int NT, int NA>
__device__ __forceinline__ void xfft_reduce(double *lds, Reduce *__restrict__ red, double ssq, MaxAbs &m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ssq += __shfl_down(ssq, off);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double ov = __shfl_down(m.v[j], off);
            const unsigned long long ol = __shfl_down(m.lin[j], off);
            const bool g = max_better(ov, ol, m.v[j], m.lin[j]);
            m.v[j]   = g ? ov : m.v[j];
            m.lin[j] = g ? ol : m.lin[j];
        }
    }
    __syncthreads();
    double *scr = lds;  // 7 doubles per wave: ssq, v[3], lin[3] (bit patterns)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        scr[wave * 7 + 0] = ssq;
        for (int j = 0; j < 3; j++) {
            scr[wave * 7 + 1 + j] = m.v[j];
            scr[wave * 7 + 4 + j] = __longlong_as_double((long long) m.lin[j]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        constexpr int NW = (NT + 63) / 64;
        double tot = 0;
        MaxAbs b;
#pragma unroll 1  // (unrolled, the 7 NW values of a 1024-thread workgroup were all loaded first: 136 spilled dwords)
        for (int i = 0; i < NW; i++) {
            tot += scr[i * 7];
            for (int j = 0; j < 3; j++) {
                const double ov = scr[i * 7 + 1 + j];
                const unsigned long long ol = (unsigned long long) __double_as_longlong(scr[i * 7 + 4 + j]);
                if (max_better(ov, ol, b.v[j], b.lin[j])) {
                    b.v[j]   = ov;
                    b.lin[j] = ol;
                }
            }
        }
        const int slot = (blockIdx.x + blockIdx.y * 7) % NSLOT;
        if (NA != 3) atomicAdd(&red->sumsq[slot], tot);
        if (NA >= 2) {
            for (int j = 0; j < 3; j++) max_commit(red, j, slot, b.v[j], b.lin[j]);
        }
    }
}

// the same through LDS only, for workgroups that are not a whole number of waves (composite sizes: N/16 threads per line); lds: 6
// doubles per thread.  A tree over the threads (log2 NT steps; a single thread scanning all NT entries cost the x stage of PPD=3456
// 60 % when this reduction was first written that way), then ONE lane commits the three axes.  All threads must call it.
template <int NT>
__device__ __forceinline__ void max_reduce_lds(double *lds, Reduce *__restrict__ red, const MaxAbs &m) {
    __syncthreads();
    for (int j = 0; j < 3; j++) {
        lds[threadIdx.x * 6 + j]     = m.v[j];
        lds[threadIdx.x * 6 + 3 + j] = __longlong_as_double((long long) m.lin[j]);
    }
    constexpr int TOP = NT <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned) (NT - 1)));  // smallest power of two >= NT
#pragma unroll 1
    for (int stride = TOP / 2; stride >= 1; stride >>= 1) {
        __syncthreads();
        const int o = (int) threadIdx.x + stride;
        if ((int) threadIdx.x < stride && o < NT) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double ov = lds[o * 6 + j], cv = lds[threadIdx.x * 6 + j];
                const unsigned long long ol = (unsigned long long) __double_as_longlong(lds[o * 6 + 3 + j]);
                const unsigned long long cl = (unsigned long long) __double_as_longlong(lds[threadIdx.x * 6 + 3 + j]);
                if (max_better(ov, ol, cv, cl)) {
                    lds[threadIdx.x * 6 + j]     = ov;
                    lds[threadIdx.x * 6 + 3 + j] = __longlong_as_double((long long) ol);
                }
            }
        }
    }
    // ONE lane commits the three axes one after the other: max_commit takes the slot's lock, and two lanes of a wave spinning on a
    // lock a third lane of the same wave holds never let it reach the release (lanes of a wave reconverge behind the loop)
    if (threadIdx.x == 0) {
#pragma unroll 1
        for (int j = 0; j < 3; j++)
            max_commit(red, j, (blockIdx.x + blockIdx.y * 7) % NSLOT, lds[j], (unsigned long long) __double_as_longlong(lds[3 + j]));
    }
}
