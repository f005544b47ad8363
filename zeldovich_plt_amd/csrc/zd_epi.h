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


// workgroup reduction of the epilogue -> one atomic per quantity into a replicated slot
template <int NT, int NA>
__device__ __forceinline__ void xfft_reduce(double *lds, Reduce *__restrict__ red, double ssq, double (&mp)[3], double (&mn)[3]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ssq += __shfl_down(ssq, off);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            mp[j] = fmax(mp[j], __shfl_down(mp[j], off));
            mn[j] = fmax(mn[j], __shfl_down(mn[j], off));
        }
    }
    __syncthreads();
    double *scr = lds;  // 7 doubles per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        scr[wave * 7 + 0] = ssq;
        for (int j = 0; j < 3; j++) {
            scr[wave * 7 + 1 + j] = mp[j];
            scr[wave * 7 + 4 + j] = mn[j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        constexpr int NW = (NT + 63) / 64;
        double tot = 0, a3[3] = {0, 0, 0}, b3[3] = {0, 0, 0};
#pragma unroll 1  // (unrolled, the 7 NW values of a 1024-thread workgroup were all loaded first: 136 spilled dwords)
        for (int i = 0; i < NW; i++) {
            tot += scr[i * 7];
            for (int j = 0; j < 3; j++) {
                a3[j] = fmax(a3[j], scr[i * 7 + 1 + j]);
                b3[j] = fmax(b3[j], scr[i * 7 + 4 + j]);
            }
        }
        const int slot = (blockIdx.x + blockIdx.y * 7) % NSLOT;
        if (NA != 3) atomicAdd(&red->sumsq[slot], tot);
        if (NA >= 2) {
            for (int j = 0; j < 3; j++) {
                atomicMax(&red->maxpos[j][slot], dbits(fabs(a3[j])));
                atomicMax(&red->maxneg[j][slot], dbits(fabs(b3[j])));
            }
        }
    }
}

