// zd_plan.h — the plan object behind the staged C ABI (include/zeldovich_hip.h), shared by zd_capi.cpp (single-rank
// stages) and zd_multi.cpp (the in-library multi-GPU driver).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/zeldovich_hip.h"
#include "zd_device.h"

struct EventPair {
    hipEvent_t a, b;
    int kind;
};

struct zd_plan {
    zd_params p;
    int rank = 0, nranks = 1;
    int N = 0, half = 0, narray = 0, R = 1, L = 0, Hq = 0, Zq = 0;
    int pack = zd::PACK_NONE;  // what the store holds (zd_device.h PACK_*)
    int npass = 1, pstep = 1;  // passes per run; planes a store plane delivers (2 with PACK_ZAPAIR)
    bool var_pending = true;   // packed stores: the next Z stage accumulates sum |D|^2
    zd::GenConst g;
    zd::GenJumps J;
    zd::JobList jobs;
    zd::StoreLayout S;
    zd::EpiConst ec;
    // field store (PACK_ZAFIELD): layout of the store, and the ring the y stage fills for the x stage
    zd::FieldLayout F;
    zd::FieldRow *d_fieldrows = nullptr;
    zd::StoreLayout SR;             // the ring as a single-rank block store of 3 arrays
    zdfft::cplx *d_ring = nullptr;
    bool dens = false;              // ZD_qdensity = 1 on the six-field store (composite grids): fields 4, 5 = D of the two residues
    bool dens_only = false;         // ZD_qdensity = 2 there: density planes only, no records
    zdfft::cplx *d_ring_dens = nullptr;  // ... and their array delta_r0 + i delta_r1 for the ring's planes
    // PLT + ZD_qdensity = 1 on the composite grids (one rank): the density planes of a pass come from a second, density-only plan at
    // stream factor 2R (its pass j = residues j, j + R = exactly the planes of this plan's pass j), run at the head of the Z stage on
    // the same store; they wait here, in delivery order, for the x stage's calls
    zd_plan *dens_sub = nullptr;
    float *d_dens_pass = nullptr;
    int ring_planes = 0;
    int64_t store_bytes_ = 0;       // bytes of the send (= receive) buffer per pass
    // device tables
    double *d_pk = nullptr;  // x | y | y2
    int *d_lut = nullptr;
    double *d_pktab = nullptr;
    double *d_fnlM = nullptr;  // f_NL: M(k) by integer |k|^2
    double *d_eig = nullptr;
    double *d_eiglines = nullptr;  // PLT with an interpolated table: (x, y)-interpolated lines of the slab being generated
    zdpcg::u128 *d_rowstate = nullptr;
    zdfft::cplx *d_twN = nullptr, *d_twL = nullptr;
    zdfft::cplx *d_twq_n = nullptr, *d_twq_l = nullptr;  // PPD = 2^a 3^b: twiddle sets of the composite transforms (lengths N, L)
    double *d_genf = nullptr;  // LDS image of k_genf
    unsigned *d_tilectr = nullptr;  // one work counter per k_genf launch of a pass
    int n_tilectr = 0, gen_max_wgs = 0;
    zd::Reduce *d_red = nullptr;
    // folded FFT inputs of one slab of half-space rows: Y[job][row][k2][x]; double-buffered so that
    // k_gen (VALU-bound) of slab s+1 runs beside k_zfft (HBM-bound) of slab s on a second stream
    // folded FFT inputs: a ring of slabs Y[job][row][k2][x].  Two slabs suffice for the gen || zfft overlap inside a
    // pass; whatever HBM the store leaves free holds more of them, so that the (VALU-bound) generator of pass p+1
    // runs ahead on its own stream while the (HBM-bound) y and x passes of pass p are still working
    std::vector<zdfft::cplx *> d_Y;
    std::vector<hipEvent_t> ev_gen, ev_fft;
    int slab_rows = 0;        // rows generated per k_gen launch
    long long next_g = 0;     // running slab number: slab g lives in ring slot g % K
    int pass_step = 1;        // distance to this rank's next pass (pass groups: zd_plan_run_passes deals passes first, first + step, ...)
    int ahead_pass = -1;      // pass whose first `ahead_n` slabs (numbers ahead_g0...) are already being generated
    long long ahead_g0 = 0;
    int ahead_n = 0;
    hipStream_t s_gen = nullptr, s_fft = nullptr;
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_pipe[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // one rank, two stores (zd_plan_run_passes): start, Z done x2, XY done x2
    bool overlap = true;
    // fused Z stage of the packed PLT store (zd_kernels_fz.hip): work items (row, first column, columns), longest first
    bool fused_z = false;
    zd::FzItem *d_fzitems = nullptr;
    unsigned n_fzitems = 0;
    int ncu = 256;
    // ZD_Version = 1 (zd_kernels_v1.hip): mt19937 streams, one per yres; accepted pairs of the slab being generated
    int v1_block = 0;               // PPD / NumBlock streams (0: version 2)
    zd::V1Stream *d_v1streams = nullptr;
    double2 *d_v1dev = nullptr;     // [slab row][z][x]
    int *d_v1err = nullptr;
    zdfft::cplx *d_phik_owned = nullptr;  // ZD_f_NL through zd_plan_create: PhiK of the phi round (zd_generate keeps its own)
    // any even PPD (zd_kernels_any.hip): Bluestein tables for the lengths L (z lines) and N (y, x lines)
    bool any = false;
    zd::AnyTab tabL = {}, tabN = {};
    zd::AnyLayout AL = {};
    zdfft::cplx *d_any[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // timing
    std::vector<EventPair> events;
    std::vector<hipEvent_t> pool;
    double kernel_ms[ZD_K_COUNT] = {};
    int64_t launches[ZD_K_COUNT] = {};
    int64_t bytes_sent = 0;  // N > 1 ranks: accounted by zd_plan_run_pass, reported and reset by zd_plan_stats
};

// internal entry points shared by zd_capi.cpp and zd_multi.cpp
extern "C" {
int zd_plan_stage_y_group(zd_plan *pl, void *d_recv, int chunk_planes, int nplanes, void *hip_stream);
int zd_plan_stage_x_group(zd_plan *pl, int residue, const void *d_recv, int chunk_planes, int64_t plane0, int64_t gplane0,
                          int64_t nplanes, void *d_records, float *d_density, void *hip_stream);
// f_NL on several ranks (zd_multi.cpp)
int zd_plan_create_phi(const zd_params *p, const zd_pk *pk, int rank, int nranks, zd_plan **out);
int zd_plan_create_phik(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank, int nranks, const void *d_phik,
                        zd_plan **out);
int zd_plan_phi_xy_group(zd_plan *pl, void *d_slot, int chunk_planes, int nplanes, double f_NL, void *hip_stream);
int zd_plan_phi_zfwd(zd_plan *pl, void *d_store, void *d_phik, void *hip_stream);
// hipEvent pair around something on `hip_stream` that is not a kernel of this file (zd_multi.cpp: the wait for an exchanged
// plane group); summed into kernel_ms[kind] by zd_plan_stats when the plan profiles
void zd_plan_tick(zd_plan *pl, int kind, void *hip_stream, int begin);
int zd_plan_stage_z_detached(zd_plan *pl, int residue, void *d_send, void *hip_stream, void *wait_event, void *done_event);
}
// hipMalloc for the large, partly written buffers (stores, rings, phi fields); NaN-filled under zd_test_poison (testing library)
hipError_t zd_store_alloc(void **p, size_t bytes);
int zd_generate_multi(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, zd_slab_cb cb, void *user,
                      zd_stats *out, int transport);
