/* zeldovich_hip.h — C ABI of the MI355X-native grid->displacements path.
 *
 * The reference (abacusorg/zeldovich-PLT, mounted at /root/reference; citations are relative to it)
 * has no plugin / FFI interface: the path sits behind three C++ call sites in main()
 * (src/zeldovich.cpp:938 Setup_FFTW, :962-971 BlockArray + ZeldovichZ, :982 ZeldovichXY, which calls
 * back WriteParticlesSlab once per z plane, :667-681).  This header is the C-ABI a maintainer binds
 * in their place — plain pointers and sizes, no C++ / torch types.  INTEGRATION.md shows the binding.
 *
 * Library: zeldovich_plt_amd/csrc/build/libzeldovich_hip.so (HIP, gfx950 only).  All entry points
 * return 0 on success and non-zero on failure after printing a message to stderr (the reference's
 * convention is message + exit(1); the CLI wrapper turns a non-zero return into exit(1)).
 */
#ifndef ZELDOVICH_HIP_H
#define ZELDOVICH_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the declarations of this header are exported */
#pragma GCC visibility push(default)

#define ZD_MAX_PPD 65536 /* include/zeldovich.h:34 — fixes the RNG addressing */

/* ICFormat -> record layout (include/output.h:19-49) */
enum { ZD_FMT_ZEL = 0, ZD_FMT_RVZEL = 1, ZD_FMT_RVDOUBLEZEL = 2, ZD_FMT_ZELSIMPLE = 3 };

/* The subset of `Parameters` (include/parameters.h:9-86) that the path reads, with the derived
 * quantities of Parameters::setup (src/parameters.cpp:172-174) already filled in. */
typedef struct zd_params {
    int64_t ppd;        /* cbrt(NP): a power of two in [32, 16384] (above 8192: ZA field store only; 8192 with PLT: its field store only, i.e. no ZD_qdensity / ZD_f_NL), or 2^a Q with a >= 5 and Q one of 3, 9, 27, 5, 15,
                         * 25, 45, 75, 125, 135, 7, 21, 35, 49 (sizes: csrc/zd_kernels_np2.hip NP2_SIZES; the composite-transform kernels: ZA and PLT, also with ZD_qdensity = 1 or 2 — PLT with a density on one rank per pass group), or ANY other even number in [8, 8192]
                         * (one rank: convolution transforms on the power-of-two engine, ~6x slower) */
    int32_t numblock;   /* ZD_NumBlock: v2 output does not depend on it; version 1: PPD / numblock random streams */
    int32_t cpd;        /* CPD: only used by the writer for ic_{z*cpd/ppd} */
    double boxsize;     /* BoxSize */
    double fundamental; /* 2 pi / boxsize */
    double nyquist;     /* pi / (boxsize/ppd) */
    double k_cutoff;    /* ZD_k_cutoff */
    double f_cluster;   /* ZD_f_cluster */
    double z_initial;   /* InitialRedshift */
    double PLT_target_z;
    int64_t seed;        /* ZD_Seed widened int -> unsigned long as src/power_spectrum.cpp:14 does */
    int32_t corner_modes;/* ZD_CornerModes */
    int32_t qdensity;    /* ZD_qdensity: 0 none, 1 also density plane, 2 density only */
    int32_t qoneslab;    /* ZD_qoneslab: -1 all, else only this z is delivered */
    int32_t qonemode;    /* ZD_qonemode */
    int32_t one_mode[3]; /* ZD_one_mode */
    int32_t qPLT;        /* ZD_qPLT */
    int32_t qPLTrescale; /* ZD_qPLT_rescale */
    int32_t icformat;    /* ZD_FMT_* */
    /* --- optional MI355X knobs (0 = let the library decide); not present in the reference --- */
    int32_t stream_factor; /* R: number of z-residue classes (a power of two; composite PPDs: any even divisor whose z lines PPD/R have a transform) */
    int32_t profile;       /* 1: bracket every kernel with hipEvents and report per-kernel ms */
    int32_t store_mode;    /* ZD_STORE_*: what the block store between the z and y passes holds (0 = best available) */
    int32_t serial_z;      /* 1: generator and z FFT on ONE stream (per-kernel timing runs); 0: two overlapped streams */
    int32_t ngpu;          /* ZD_NumGPU: GPUs of this node that zd_generate / the CLI drive (0 or 1 = one) */
    int32_t exchange_planes; /* store planes per exchange group between ranks (0 = ~4 GB ring slots) */
    /* --- local primordial non-Gaussianity (include/parameters.h:56-58); f_NL = 0 disables the path --- */
    double f_NL, n_s, Omega_M;
    /* --- ZD_Version (include/parameters.h:67-72): 0 or 2 = the pcg64 counter streams; 1 = the legacy phases: one
     * gsl_rng_mt19937 per yres (seed + yres) with rejection sampling (src/power_spectrum.cpp:18-25,310-332), which makes
     * the field depend on `numblock` — pass the value AFTER the reader's adjustment numblock = numblock * k_cutoff + .5
     * (src/parameters.cpp:129-141; zd_read_params does it) --- */
    int32_t version;
    int32_t pass_groups;   /* ZD_PassGroups, with ngpu > 1: the GPUs work as `pass_groups` independent groups of ngpu / pass_groups
                            * ranks; group j takes the residue passes j, j + pass_groups, ... and nothing travels between
                            * groups (the z-residue classes of the streaming are independent partitions of the output).
                            * Inside a group the rows / planes are sharded with the exchange described below.  0 = automatic:
                            * one GPU per group while the job has at least ngpu passes (2 and 4 GPUs at PPD = 4096: no
                            * exchange at all, where a pairwise exchange would be bound by ONE xGMI link), else one group
                            * of all GPUs (the all-to-all of 8 GPUs, every link of the mesh busy) */
} zd_params;

/* zd_params.store_mode */
enum {
    ZD_STORE_AUTO = 0,
    ZD_STORE_REFERENCE = 1, /* the reference's 1 / 2 / 4 complex arrays (density transformed; include/block_array.h:26-35) */
    ZD_STORE_PACKED = 2,    /* 3 arrays without the density field: ZA two z-residues per pass, PLT qx+i vx | qy+i qz | vy+i vz */
    ZD_STORE_FIELDS = 3     /* ZA only: the two potentials E = sum D/k^2, Z = sum kz D/k^2 of two z-residues for the
                             * half-space rows (no Hermitian twins), zero columns not stored; the y pass derives the
                             * displacement arrays plane by plane */
};

/* PowerSpectrum state after InitFromFile/InitFromPowerLaw + Normalize (src/power_spectrum.cpp:130-223).
 * Tables are the (ln k, ln P, y'') arrays of SplineFunction (include/spline_function.h). */
typedef struct zd_pk {
    int32_t n;
    const double *x, *y, *y2;
    double normalization;
    double Pk_smooth2;
    int32_t fixed_power; /* ZD_qPk_fix_to_mean */
    int32_t is_powerlaw;
    double powerlaw_index;
    double kmax; /* largest tabulated k (extrapolation warning only) */
    double kmin; /* smallest positive tabulated k (1e-4 for a power law): anchors primordial_norm, f_NL only */
} zd_pk;

/* Names of the kernels in timing arrays */
enum { ZD_K_GEN = 0, ZD_K_ZFFT = 1, ZD_K_YFFT = 2, ZD_K_XFFT = 3,
       ZD_K_ZSTAGE = 4, /* the Z stage as one unit: first generator launch .. last z FFT of a pass (the two kernels overlap
                         * on two streams, so their own spans include each other) */
       ZD_K_XWAIT = 5,  /* N > 1 ranks: time the compute stream stood waiting for an exchanged plane group to arrive */
       ZD_K_COUNT = 6 };

typedef struct zd_stats {
    double max_disp[3];      /* output.cpp:28: signed value of the largest |displacement| per axis (x,y,z) */
    double density_variance; /* output.cpp:30: sum of dens^2 over delivered planes */
    double seconds_total;    /* wall time of the grid->displacements interval (host clock, synced) */
    double kernel_ms[ZD_K_COUNT];      /* summed hipEvent time per kernel (profile=1) */
    int64_t kernel_launches[ZD_K_COUNT];
    int64_t bytes_intermediate; /* size of the z-FFT'd block store held in HBM per residue pass */
    int32_t stream_factor;      /* R actually used */
    int32_t modes_cached;       /* 1 if the Gaussian mode amplitudes were kept in HBM across passes */
    int64_t bytes_sent;         /* N > 1 ranks: bytes sent to OTHER ranks since the last zd_plan_stats / by this zd_generate call
                                 * (zd_generate: summed over the ranks) */
    int64_t max_disp_index[3];  /* lattice site (z * ppd + y) * ppd + x of max_disp[j]: the FIRST site in the reference's (z, y, x)
                                 * loop order that holds the largest |displacement| (output.cpp:190-193 compares with a strict >);
                                 * -1 if the field is identically zero */
} zd_stats;

/* Replacement for the per-plane callback WriteParticlesSlab (src/output.cpp:41-234).
 *   z          plane index (reference calls in increasing z; with stream_factor R > 1 planes arrive in
 *              residue order r, r+R, ... and `z` tells the consumer where they belong)
 *   n_records  ppd*ppd
 *   records    HOST pointer to n_records packed records (ICFormat layout), valid during the call only
 *   density    HOST pointer to n_records float32 densities, or NULL unless qdensity != 0
 * Return non-zero to abort. */
typedef int (*zd_slab_cb)(void *user, int64_t z, int64_t n_records, const void *records,
                          const float *density);

/* One call = ZeldovichZ + ZeldovichXY (src/zeldovich.cpp:517-695) on one GPU, or on p->ngpu GPUs of this node (one host thread
 * per GPU inside the library, the block exchange over RCCL / xGMI; the callback still comes from one thread at a time).
 *   eig / eig_ppd: PLT eigenmode table as loaded by load_eigmodes (src/zeldovich.cpp:794-830),
 *                  [eig_ppd][eig_ppd][eig_ppd/2+1][4] doubles; NULL/0 unless qPLT.
 *   cb may be NULL: planes are then produced in HBM and dropped (benchmark sink). */
int zd_generate(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, zd_slab_cb cb,
                void *user, zd_stats *out);

/* Smallest stream factor R (a power of two; composite PPDs: any even divisor with a supported z length) whose block store of one pass (+ the y->x ring of the field store, + the exchange
 * ring for nranks > 1) fits in budget_bytes; -1 if none. */
int zd_choose_stream_factor(const zd_params *p, int nranks, int64_t budget_bytes);
/* How `ngpu` GPUs share the job (zd_params.pass_groups, 0 = automatic): *groups independent groups of ngpu / *groups ranks
 * and the stream factor *stream_factor whose number of passes is a multiple of *groups (budget_bytes: free HBM per rank, as
 * for zd_choose_stream_factor).  zd_generate applies it; one-process-per-GPU drivers call it so that every rank arrives at
 * the same split.  Returns non-zero if nothing fits. */
int zd_choose_pass_groups(const zd_params *p, int ngpu, int64_t budget_bytes, int32_t *groups, int32_t *stream_factor);
/* The same with a MEASURED link rate (zd_comm_probe: GB/s one link carries per direction while every link of a GPU works; <= 0 = not
 * measured, i.e. zd_choose_pass_groups).  Where the automatic choice would be one GPU per pass group (no exchange), the time of a
 * step is estimated for both splits from the per-particle unit times of one MI355X (generation, z FFT, y + x stages)
 * and the exchange's bytes per link at that rate, and the single group with the all-to-all (block store transposed between the z and
 * the y / x passes, the reference's StoreBlock / LoadBlock: src/block_array.cpp:387-414,466-504) is taken when it comes out faster —
 * about 45 GB/s per link at PPD = 4096 on 8 GPUs.  est_seconds (may be NULL): [0] pass groups, [1] all-to-all. */
int zd_choose_pass_groups_measured(const zd_params *p, int ngpu, int64_t budget_bytes, double link_GBps, int32_t *groups,
                                   int32_t *stream_factor, double *est_seconds);

/* ---- staged API (device pointers) for one-process-per-GPU drivers and for tests ---------------
 * Rank `rank` of `nranks` (a power of two) owns the half-space rows ky = rank, rank + nranks, ... (H = ppd/2/nranks
 * of them; cyclic, because the rows near ky = 0 carry most of the non-zero modes), plus their Hermitian twins,
 * during the Z stage, and z planes [rank*Zq, (rank+1)*Zq) of every residue
 * pass (Zq = ppd/R/nranks) during the XY stage.  Between the two, the caller exchanges equal
 * chunks (all-to-all): chunk d of the send buffer goes to rank d and is received as chunk `rank`…
 * of the receive buffer (exactly torch.distributed.all_to_all_single / ncclAllToAll semantics).
 * With nranks == 1 the send buffer IS the receive buffer.
 * ZD_f_NL != 0 (one rank): zd_plan_create runs the phi round of the reference (src/zeldovich.cpp:945-960) once and the plan
 * keeps PhiK; its Z stages then read D = PhiK * M. */
typedef struct zd_plan zd_plan;

int zd_plan_create(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, int rank,
                   int nranks, zd_plan **out);
void zd_plan_destroy(zd_plan *plan);

int32_t zd_plan_narray(const zd_plan *plan);       /* arrays of the store: 1, 2 or 4 (src/zeldovich.cpp:871-876), or 3:
                                                     * without ZD_qdensity the density field is not transformed and the
                                                     * remaining 3 (ZA) / 6 (PLT) real fields are packed into 3 arrays —
                                                     * ZA packs TWO z-residues per pass (qy+i qz of each, qx_r0 + i qx_r1),
                                                     * PLT packs qx+i vx | qy+i qz | vy+i vz; density_variance then comes
                                                     * from sum |D(k)|^2 (Parseval) */
int32_t zd_plan_store_mode(const zd_plan *plan);   /* ZD_STORE_REFERENCE / _PACKED / _FIELDS: what the store of this plan holds */
int32_t zd_plan_stream_factor(const zd_plan *plan);/* R: z-residue classes */
int32_t zd_plan_passes(const zd_plan *plan);       /* passes per run: R, or R/2 when a pass carries two residues */
int32_t zd_plan_plane_step(const zd_plan *plan);   /* 1, or 2 when a pass carries two residues: stage_x plane ranges
                                                     * must be multiples of it */
int32_t zd_plan_record_size(const zd_plan *plan);  /* bytes per particle record */
int64_t zd_plan_exchange_bytes(const zd_plan *plan); /* bytes of the send (= receive) buffer per pass */
int64_t zd_plan_local_planes(const zd_plan *plan);   /* Zq: z planes this rank finishes per pass */
int64_t zd_plan_plane_z(const zd_plan *plan, int residue, int64_t local_plane); /* global z */

/* Z stage for residue pass `residue`: mode generation (or cache reuse) + folded z FFT for the rows
 * this rank owns, written into `d_send` (device pointer, zd_plan_exchange_bytes). */
int zd_plan_stage_z(zd_plan *plan, int residue, void *d_send, void *hip_stream);

/* XY stage: in-place y FFT on `d_recv`, then x FFT + particle epilogue for local planes
 * [plane0, plane0+nplanes).  d_records: nplanes*ppd*ppd records; d_density: float32 or NULL.
 * Call zd_plan_stage_x ONCE per plane and pass: for the PPDs that run as convolutions (neither 2^a nor 2^a 3^b) the x transform
 * of a plane is done in place in `d_recv`. */
int zd_plan_stage_y(zd_plan *plan, void *d_recv, void *hip_stream);
int zd_plan_stage_x(zd_plan *plan, int residue, const void *d_recv, int64_t plane0, int64_t nplanes,
                    void *d_records, float *d_density, void *hip_stream);

/* ---- N > 1 ranks: exchange + pipelined XY stages inside the library --------------------------------
 * Replaces the per-block StoreBlock / LoadBlock traffic of src/zeldovich.cpp:583-587,634-637 between ZeldovichZ and
 * ZeldovichXY.  The send store is [destination rank][plane]...: plane groups of every chunk are contiguous, group j+1
 * travels (RCCL grouped ncclSend/ncclRecv over xGMI) while the y and x stages of group j run.  zd_generate with
 * zd_params.ngpu > 1 (`ZD_NumGPU` in the parameter file) drives all of this with one host thread per GPU; the entry
 * points below serve one-process-per-GPU drivers (bench.py under torch.distributed.run). */
typedef struct zd_comm zd_comm;
int zd_comm_unique_id(void *id128);  /* rank 0: 128-byte RCCL id, to be made known to all ranks by the launcher */
int zd_comm_create(int rank, int nranks, const void *id128, zd_comm **out); /* ncclCommInitRank on the current device */
void zd_comm_destroy(zd_comm *comm);
/* A rank that cannot go on (failed allocation, consumer error, ...) calls this before returning its error: ncclCommAbort on
 * its communicator, so that the send / receive kernels it has already queued drain instead of spinning on peers for ever.
 * zd_plan_run_pass does it by itself on every failure path; the launcher is expected to stop the other ranks when one
 * process exits non-zero (torch.distributed.run does). */
void zd_comm_abort(zd_comm *comm);
/* bytes this rank has sent to / received from other ranks since the communicator was created (or since the last call with
 * reset != 0) */
void zd_comm_traffic(zd_comm *comm, int64_t *bytes_sent, int64_t *bytes_received, int reset);
/* timed probe of the links: bytes_per_peer to and from every peer at once (grouped send / receive), one warm-up + reps repetitions;
 * *GBps_per_peer = what one link carries per direction meanwhile (0: no peers).  Every rank of the communicator calls it. */
int zd_comm_probe(zd_comm *comm, int64_t bytes_per_peer, int32_t reps, double *GBps_per_peer);
/* bytes of the two-slot receive ring zd_plan_run_pass allocates (0 for one rank) and the planes per exchange group */
int64_t zd_plan_ring_bytes(const zd_plan *plan, int32_t *group_planes);
/* consumer of finished planes: `nplanes` delivered planes starting at local plane `first_local_plane` of the pass lie
 * in d_records, produced by work still queued on hip_stream (order yourself after it, or synchronise) */
typedef int (*zd_group_cb)(void *user, int64_t first_local_plane, int64_t nplanes, const void *d_records, void *hip_stream);
/* One residue pass of this rank: Z stage into d_store (zd_plan_exchange_bytes), exchange and XY stages plane group by
 * plane group through d_records (room for rec_planes planes, a multiple of zd_plan_plane_step).  comm == NULL for one
 * rank.  cb may be NULL (benchmark sink). */
int zd_plan_run_pass(zd_plan *plan, zd_comm *comm, int pass, void *d_store, void *d_records, int64_t rec_planes,
                     zd_group_cb cb, void *user, void *hip_stream);

/* The passes first, first + step, ... of this rank (step = number of pass groups, first = this rank's group) in one call.
 * d_store2 (optional, as large as d_store): with a communicator of several ranks the passes are then PIPELINED — the Z stage of
 * the next pass runs into the other store while the planes of the current pass are exchanged and transformed; a store is
 * rewritten only after its sends have completed.  With ONE rank (comm == NULL) and a second store the Z stage of the next pass is
 * issued beside the y / x stages of the current one (measured slower than one store at PPD=4096 on an MI355X — DESIGN.md §8 —
 * and therefore not what zd_generate does; kept for A/B runs, bench.py --two-stores).  cb additionally receives the pass. */
typedef int (*zd_pass_cb)(void *user, int pass, int64_t first_local_plane, int64_t nplanes, const void *d_records, void *hip_stream);
int zd_plan_run_passes(zd_plan *plan, zd_comm *comm, int first, int step, void *d_store, void *d_store2, void *d_records,
                       int64_t rec_planes, zd_pass_cb cb, void *user, void *hip_stream);

/* Fetch + reset the device-side reductions (max_disp, density_variance) and kernel timers. Syncs. */
int zd_plan_stats(zd_plan *plan, zd_stats *out);

/* ---- host-side helpers mirroring the reference's setup code (no GPU needed) -------------------- */
/* Parameters(file) + setup(): src/parameters.cpp:11-197.  Fills zd_params; strings via out buffers. */
typedef struct zd_param_strings {
    char Pk_filename[1024];
    char output_dir[1024];
    char density_filename[1024];
    char PLT_filename[1024];
    char ICFormat[64];
    double Pk_scale, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth, Pk_powerlaw_index;
    int32_t qPk_fix_to_mean;
    int32_t version;
    double f_NL, n_s, Omega_M;
    int64_t np;
} zd_param_strings;
int zd_params_from_file(const char *path, zd_params *p, zd_param_strings *s);

/* PowerSpectrum: InitFromFile / InitFromPowerLaw + Normalize.  The handle owns the tables that
 * zd_pk points into. */
typedef struct zd_pk_handle zd_pk_handle;
int zd_pk_create_from_file(const char *path, double Pk_scale, double Pk_norm, double Pk_sigma,
                           double Pk_sigma_ratio, double Pk_smooth, int fix_to_mean, double boxsize,
                           zd_pk_handle **h, zd_pk *pk);
int zd_pk_create_powerlaw(double index, double Pk_norm, double Pk_sigma, double Pk_sigma_ratio,
                          double Pk_smooth, int fix_to_mean, double boxsize, zd_pk_handle **h, zd_pk *pk);
double zd_pk_power(const zd_pk *pk, double k);  /* PowerSpectrum::power, src/power_spectrum.cpp:225 */
double zd_pk_sigmaR(const zd_pk *pk, double R); /* PowerSpectrum::sigmaR, src/power_spectrum.cpp:60 */
void zd_pk_destroy(zd_pk_handle *h);

/* load_eigmodes: src/zeldovich.cpp:794-830.  Caller frees with zd_free. */
int zd_load_eigmodes(const char *path, double **eig, int64_t *eig_ppd);
void zd_free(void *p);

/* ---- diagnostics ------------------------------------------------------------------------------
 * Which kernel variants this process has launched so far, and how often: one line "count<TAB>line<TAB>launcher" per launch
 * site of the library, the launcher named with its template arguments (transform length, elements per thread, tile shape,
 * packing).  Writes at most cap - 1 bytes + a terminating 0 into buf (may be NULL) and returns the size a complete report
 * needs.  (The reference logs its FFTW plan sizes at start-up, src/zeldovich.cpp:39-135; here the variant depends on PPD,
 * store and stream factor, and the test-suite uses this report to prove that every variant it ships has been exercised.) */
int64_t zd_dispatch_report(char *buf, int64_t cap);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
