// zd_testing.h — entry points that exist only in the -DZD_TESTING library (make testing ->
// build/libzeldovich_hip_testing.so): device test hooks and the in-process emulation of the RCCL calls.  They are test
// scaffolding, not part of the product C ABI (include/zeldovich_hip.h); tests/ reach them through
// zeldovich_plt_amd.api.load_testing_library().
#pragma once
#include "../../include/zeldovich_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
/* ---- device test hooks (each needs a GPU) -------------------- */
/* n counter-addressed draws: out[2*i], out[2*i+1] = the two uint64 of mode (kx,ky,kz)[i] */
int zd_test_draws(int64_t seed, int64_t n, const int32_t *kxyz, uint64_t *out);
/* Gaussian amplitudes D(k) for the same mode list (cgauss<2>, src/power_spectrum.cpp:338-359) */
int zd_test_modes(const zd_params *p, const zd_pk *pk, int64_t n, const int32_t *kxyz, double *D);
/* the same through the arithmetic the production generator k_genf uses (LDS-table ln / exp / sincos / spline segments,
 * integer zero rule, Newton reciprocal): out[3*i] = {Re D, Im D, fundamental / |k|^2}; ky >= 0 */
int zd_test_modes_table(const zd_params *p, const zd_pk *pk, int64_t n, const int32_t *kxyz, double *out);
/* zd_generate with p->ngpu ranks as threads on the visible GPU(s), running the RCCL branch of the exchange code (buffer
 * offsets, grouped send / receive order, stream and event ordering) on an in-process emulation of the ncclSend / ncclRecv /
 * ncclGroup calls — real RCCL refuses two ranks on one device, and test boxes have one */
int zd_test_generate_loopback(const zd_params *p, const zd_pk *pk, const double *eig, int64_t eig_ppd, zd_slab_cb cb, void *user,
                              zd_stats *out);
/* ZD_Version = 1: the first 624 * nblocks words of gsl_rng_mt19937 seeded with `seed`, from the workgroup-parallel
 * regeneration the stream kernel uses (src/power_spectrum.cpp:18-25) */
int zd_test_v1_words(int64_t seed, int32_t nblocks, uint32_t *out);
/* batch of `lines` independent length-n inverse FFTs, host in/out [lines][n] complex double;
 * axis_kind 0: the contiguous-line kernel path (x pass), 1: the strided-line path (y/z passes) */
int zd_test_fft(int32_t n, int64_t lines, int32_t axis_kind, const double *in, double *out);
/* on != 0: every store / exchange ring / phi field the library allocates from now on starts out as NaN bytes */
void zd_test_poison(int on);
/* rank >= 0: that rank of the thread-per-GPU driver (zd_generate with ngpu > 1, zd_test_generate_loopback) fails BEFORE its first
 * pass — its peers are then blocked inside their first grouped send / receive, which only an abort ends; -1: nobody */
void zd_test_fail_rank(int rank);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
