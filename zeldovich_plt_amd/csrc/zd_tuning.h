// zd_tuning.h — extra entry points of the -DZD_TUNING library (make tuning): measurement harnesses used by scripts/,
// never part of the product C ABI (include/zeldovich_hip.h).
#pragma once
#include <stdint.h>
extern "C" {
#pragma GCC visibility push(default)
/* tuning harness: ms per launch of y-pass tile variant `variant` on a synthetic store of nplanes planes */
int zd_test_yfft_variant(int32_t n, int32_t variant, int32_t narray, int32_t nplanes, int32_t tiled, int32_t reps,
                         double *ms_per_launch);
/* device copy bandwidth probe: bytes moved per second by a 16 B/lane streaming copy of `bytes` */
int zd_test_copy_bw(int64_t bytes, int32_t reps, double *gbps);

#pragma GCC visibility pop
}
