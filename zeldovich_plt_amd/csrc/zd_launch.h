// zd_launch.h — host-callable launchers implemented in zd_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "zd_device.h"

// Dispatch diagnostics (zd_dispatch_report, include/zeldovich_hip.h): every launch site counts its launches under the name of
// the launcher instantiation it sits in (template arguments included) — which kernel variant a configuration really took.
// One relaxed atomic increment per launch; a site registers itself in a lock-free list the first time it is reached.
#include <atomic>
namespace zd {
struct DispatchSite {
    const char *func;
    int line;
    std::atomic<long long> count{0};
    DispatchSite *next = nullptr;
    DispatchSite(const char *f, int l);
};
}  // namespace zd

#define ZD_LAUNCH_CHECK()                                                                       \
    do {                                                                                        \
        static zd::DispatchSite site__(__PRETTY_FUNCTION__, __LINE__);                          \
        site__.count.fetch_add(1, std::memory_order_relaxed);                                   \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "zeldovich_hip: launch failed at %s:%d: %s\n", __FILE__, __LINE__,  \
                    hipGetErrorString(e__));                                                    \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, device): ZD_NumGPU runs one host thread per device through
// the same launchers, so what has been set is kept per device — a process-wide flag would leave
// every device but the first with the 64 KB default and the launches of the large-LDS kernels would fail there.
#if defined(__HIPCC__)
#include <atomic>
template <auto Kernel>
static inline void set_dyn_lds(size_t bytes) {
    // largest size set so far per device: a kernel whose dynamic LDS depends on run-time arguments gets the attribute raised when
    // a later launch needs more than the first one did
    static std::atomic<size_t> have[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::atomic<size_t> &h = have[dev & 63];
    if (bytes == 0 || h.load(std::memory_order_acquire) >= bytes) return;
    hipFuncSetAttribute((const void *) Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
    size_t cur = h.load(std::memory_order_relaxed);
    while (cur < bytes && !h.compare_exchange_weak(cur, bytes, std::memory_order_release, std::memory_order_relaxed)) {
    }
}
#endif

namespace zd {
constexpr int GEN_BX = 256;  // threads (consecutive x) per generator workgroup
constexpr int GEN_ZR = 16;   // z rows walked by one generator thread

// tile_ctr: zeroed device counter for this launch (k_genf pulls its tiles from it; NULL -> general kernel only);
// max_wgs: size of k_genf's persistent grid
// residue2: second z-residue of the pass (PACK_ZAPAIR only)
int launch_gen(const GenConst &g, const GenJumps &J, const JobList &jobs, const StoreLayout &S, int ky0, int nky, int L,
               int residue, int residue2, const void *twN, void *Y, unsigned *tile_ctr, int max_wgs, hipStream_t st);
int launch_pk_table(const GenConst &g, int n, void *tab, hipStream_t st);
int launch_eig_lines(const GenConst &g, int ky0, int ky_stride, int nrows, void *lines, hipStream_t st);
int launch_test_modes(const GenConst &g, long long n, const int *kxyz, uint64_t *draws, double *D, hipStream_t st);
int launch_test_modes_table(const GenConst &g, long long n, const int *kxyz, double *out, hipStream_t st);
int launch_zfft(int L, const JobList &jobs, const StoreLayout &S, int ky0, int kyloc0, int nky, int Zq, const void *Y,
                const void *twL, void *out, hipStream_t st);
int zfft_tile_width(int L);
int zfft_fields_tile_columns(int L);
int launch_zfft_fields(int L, const FieldLayout &F, const StoreLayout &S, int ky0, int kyloc0, int nky, const void *Y,
                       const void *twL, void *out, hipStream_t st);
int launch_yfft_fields(const FieldLayout &F, const StoreLayout &S, const void *tw, const void *store, int plane0,
                       int nplanes, int ring_pitch, void *ring, hipStream_t st);
int launch_yfft(const StoreLayout &S, int nplanes, const void *tw, void *data, hipStream_t st);
int launch_xfft(const StoreLayout &S, const EpiConst &ec, const void *tw, const void *data, int plane0, int nplanes,
                int z_first, int z_step, void *records, float *density, Reduce *red, hipStream_t st);
int launch_yfft_variant(int variant, const StoreLayout &S, int nplanes, const void *tw, void *data, hipStream_t st);
int launch_fnl_table(const GenConst &g, int n, void *tab, hipStream_t st);
int launch_fnl_stage(int which, const StoreLayout &S, double f_NL, const void *tw, void *data, void *phik, int nplanes, int lZq,
                     hipStream_t st);
int launch_test_fft(int n, int kind, const void *tw, const void *in, void *out, long long lines, hipStream_t st);
int test_fft_tile_width(int n);
// ---- fused generator + z FFT of the packed PLT store (zd_kernels_fz.hip) ----
bool genz_plt_supported(int N, int L);
int launch_genz_plt(const GenConst &g, const StoreLayout &S, int ky0, int L, int residue, unsigned nitems, const FzItem *items,
                    const void *twN, const void *twL, void *out, unsigned *ctr, int ncu, hipStream_t st);
// ---- PPD = 2^a 3^b (zd_kernels_np2.hip) ----
bool np2_split(int n, int *P, int *Q);
int test_fftq_tile_width(int n);
bool np2_supported_ppd(int N);
bool np2_supported_zlen(int L);
int zfft_fields_np2_columns(int L);
int launch_zfft_fields_np2(int L, const FieldLayout &F, const StoreLayout &S, int ky0, int kyloc0, int nky, const void *Y,
                           const void *tw, void *out, hipStream_t st);
int launch_yfft_fields_np2(const FieldLayout &F, const StoreLayout &S, const void *tw, const void *store, int plane0, int nplanes,
                           int ring_pitch, void *ring, int dens, hipStream_t st);
int launch_xdens_np2(int N, const void *tw, const void *dring, int ring_pitch, int nplanes, float *density, hipStream_t st);
int launch_xfft_np2(int N, const EpiConst &ec, const void *tw, const void *ring, int ring_pitch, int nplanes, int z_first, int z_step,
                    void *records, Reduce *red, hipStream_t st);
int launch_test_fftq(int n, int kind, const void *twP, const void *twN, const void *twQ, const void *in, void *out, long long lines,
                     hipStream_t st);
// ---- any even PPD (zd_kernels_any.hip) ----
int any_engine_size(int n);
int launch_any_cols(const AnyTab &tb, void *data, long long batch_stride, long long point_stride, int ncols, int nbatch, int zero_point,
                    hipStream_t st);
int launch_any_lines(const AnyTab &tb, void *data, long long pitch, long long nlines, hipStream_t st);
int launch_any_scatter(const JobList &jobs, const AnyLayout &A, int ky0, int nky, int L, const void *Y, void *store, hipStream_t st);
int launch_any_emit(const AnyLayout &A, const EpiConst &ec, const void *store, int plane0, int nplanes, int z_first, int z_step, void *records,
                    float *density, Reduce *red, hipStream_t st);
int launch_any_phi_nl(const AnyLayout &A, double f_NL, void *store, hipStream_t st);
int launch_any_phik(const AnyLayout &A, const void *store, void *phik, hipStream_t st);
// ---- ZD_Version = 1 streams (zd_kernels_v1.hip) ----
int launch_v1_seed(unsigned long long seed, int block, V1Stream *streams, hipStream_t st);
int launch_v1_draw(const GenConst &g, int block, int ky0, int ky_stride, int nrows, V1Stream *streams, void *dev, int *err,
                   hipStream_t st);
int launch_test_v1_words(V1Stream *streams, int nblocks, uint32_t *out, hipStream_t st);
int launch_copy16(const void *in, void *out, long long n16, hipStream_t st);
}  // namespace zd

extern "C" int zdk_upload_bit_table(const zdpcg::BitTable *host);
extern "C" int zdk_upload_bit_table_fz(const zdpcg::BitTable *host);
