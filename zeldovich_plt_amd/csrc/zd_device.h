// zd_device.h — launch-constant structures shared by the host launcher and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "zd_fft.h"
#include "zd_pcg.h"

namespace zd {

// Everything the mode generator needs (restates the per-run scalars of LoadPlane,
// src/zeldovich.cpp:299-320, and the PowerSpectrum members used by power()/cgauss<2>).
struct GenConst {
    int N, half;
    int kmax;          // int(double(N/2)/k_cutoff + .5)           zeldovich.cpp:350
    int corner_modes;  //                                          zeldovich.cpp:353
    int qonemode, one_mode[3];
    double fundamental, fundamental2, k2_cutoff;
    // PowerSpectrum
    int pk_n, fixed_power, is_powerlaw;
    const double *pk_x, *pk_y, *pk_y2;
    double pk_norm, pk_smooth2, powerlaw_index;
    // PLT
    int qPLT, qPLTrescale;
    double f_cluster, target_f, ln_growth_ratio;  // log(a_NL/a0)
    const double *eig;
    long long eig_ppd;
    // RNG: state at the start of each ky plane (== reference v2rng[ky], power_spectrum.cpp:30-36)
    const zdpcg::u128 *row_state;
};

// Affine maps used by the generator's z-walk (set per launch geometry)
struct GenJumps {
    zdpcg::Affine jz;   // next row, same x:            advance 2*65536 - 1      (state kept one ahead)
    zdpcg::Affine jzw;  // next row across z = N/2:     advance 2*65536*(1 + 65536 - N) - 1
};

// Addressing of the z-transformed block store ("BlockArray", include/block_array.h:26-35, re-laid
// out for the GPU): element (plane zl, array a, row ky, column kx) of chunk `c` lives at
//      c*chunk_stride + zl*z_stride + a*a_stride + loc(ky)*N + kx          [complex doubles]
// where rows are grouped by the rank that generated them: rank g holds half-space rows
// [g*Hq,(g+1)*Hq) at loc 0..Hq-1 and their Hermitian twins at loc Hq..2Hq-1 — the reference's
// "displaced twin" storage (zeldovich.cpp:453-466, block_array.cpp:487-491); the unused twin slot of
// ky = 0 is the Nyquist row ky = N/2, which is never read (treated as zero, zeldovich.cpp:644-650).
struct StoreLayout {
    int N, half, Hq, narray;
    long long chunk_stride, z_stride, a_stride;
};

ZD_HD long long row_offset(const StoreLayout &L, int ky) {  // chunk (source rank) + loc for row ky
    int kyh, tw;
    if (ky < L.half) {
        kyh = ky;
        tw  = 0;
    } else if (ky == L.half) {
        kyh = 0;
        tw  = 1;  // Nyquist row -> spare twin slot of ky = 0
    } else {
        kyh = L.N - ky;
        tw  = 1;
    }
    const int src = kyh / L.Hq, loc = kyh - src * L.Hq + tw * L.Hq;
    return (long long) src * L.chunk_stride + (long long) loc * L.N;
}

// Jobs of the z stage: which real-linear combination of the mode's fields is transformed and where
// the result goes (see DESIGN.md "Hermitian pairing as FFT jobs").
enum JobKind {
    JOB_A_SELF  = 0,  // (1 - sx) D                -> array 0, row ky
    JOB_A_TWIN  = 1,  // conj FFT[(1 + sx) D]      -> array 0, row N-ky, column N-kx
    JOB_B_SELF  = 2,  // (-sz + i sy) D            -> array 1
    JOB_B_TWIN  = 3,  // conj FFT[(sz + i sy) D]   -> array 1 twin
    JOB_C_BOTH  = 4,  // (-f sx) D -> array 2 self; -conj -> array 2 twin
    JOB_D_SELF  = 5,  // f(-sz + i sy) D           -> array 3
    JOB_D_TWIN  = 6,  // conj FFT[f(sz + i sy) D]  -> array 3 twin
    JOB_DENS    = 7   // D -> array 0 self; conj -> twin         (qdensity == 2)
};
struct JobList {
    int n;
    int kind[8];
};

struct EpiConst {
    int N, narray, icformat, recsize;
    int qPLT, qdensity;
    double vnorm;  // (sqrt(1+24 f_cluster)-1)/4 without PLT, 1 with   output.cpp:78-82
};

// device-side reductions (output.cpp:28-30,190-197): NSLOT replicated accumulators
constexpr int NSLOT = 64;
struct Reduce {
    double sumsq[NSLOT];
    unsigned long long maxpos[3][NSLOT];  // bit pattern of max(+v)
    unsigned long long maxneg[3][NSLOT];  // bit pattern of max(-v)
};

}  // namespace zd
