// TEST INFRASTRUCTURE ONLY (oracle/_ref): thin extern "C" driver around the REFERENCE's own BlockArray, RAM mode
// (no -DDISK), linked with /root/reference/src/block_array.cpp and src/STimer.cc compiled where they lie (oracle/Makefile,
// target `ref`; needs only the image's header-only fmt under torch/include).  Nothing here restates reference code: it
// fills slabs, calls BlockArray::StoreBlock / LoadBlock (src/block_array.cpp:387-414,466-504) and copies out what they
// produced.  Used by tests/golden/make_golden.py to generate tests/golden/blockarray_kat.json, which pins
// oracle/zd_oracle.c store_block/load_block and the layout `arr[zblock][yblock][a][zres][yres][x]`
// (include/block_array.h:33-34).
#include <cstdint>
#include <cstring>
// `arr` sits in the class's leading (default-private) section: this driver TU reads it to dump the global layout, so
// it parses the class as a struct; every header block_array.h needs is included normally first, and the reference's
// own TU (src/block_array.cpp) is compiled unchanged.
#include <complex>
#include <filesystem>
#include <mutex>
#include "STimer.h"
#include "zeldovich.h"
#define class struct
#include "block_array.h"
#undef class

extern "C" {

// slabs_in : [numblock yblocks] slabs of the z stage, each [block yres][narray][ppd z][ppd x] complex (BLK_AYZX)
// arr_out  : the whole BlockArray after all StoreBlock calls, ppd^3 * narray complex
// slabs_out: [numblock zblocks] slabs of the xy stage, each [block zres][narray][ppd y][ppd x] complex (BLK_AZYX),
//            pre-filled with `fill` so that untouched rows (the Nyquist row, block_array.cpp:487-494) are visible
int ref_blockarray_roundtrip(int ppd, int numblock, int narray, const double *slabs_in, double *arr_out,
                             double *slabs_out, double fill) {
    BlockArray ba(ppd, numblock, narray, fs::path("."), 0, 0, -1);
    const int64_t block = ppd / numblock, slab_c = (int64_t) block * narray * ppd * ppd;
    for (int yblock = 0; yblock < numblock; yblock++)
        for (int zblock = 0; zblock < numblock; zblock++)
            ba.StoreBlock(yblock, zblock, (Complx *) (slabs_in + 2 * slab_c * yblock));
    memcpy(arr_out, ba.arr, sizeof(Complx) * (size_t) ppd * ppd * ppd * narray);
    for (int64_t i = 0; i < 2 * slab_c * numblock; i++) slabs_out[i] = fill;
    for (int zblock = 0; zblock < numblock; zblock++)
        for (int yblock = 0; yblock < numblock; yblock++)
            ba.LoadBlock(yblock, zblock, (Complx *) (slabs_out + 2 * slab_c * zblock));
    return 0;
}
}
