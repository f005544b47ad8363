// zd_device.h — launch-constant structures shared by the host launcher and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "zd_fft.h"
#include "zd_pcg.h"
#if !defined(__HIPCC__)
struct double2 { double x, y; };
#endif

// Ablation / tuning branches (GenConst::ablate, StoreLayout::prune bits >= 3, StoreLayout::nt) are compiled only
// into the -DZD_TUNING library; in the product they fold to `false`.
#ifdef ZD_TUNING
#define ZD_TUNE(x) ((x) != 0)
#else
#define ZD_TUNE(x) (false)
#endif

namespace zd {

constexpr int PK_LUT = 2048;

// Everything the mode generator needs (restates the per-run scalars of LoadPlane,
// src/zeldovich.cpp:299-320, and the PowerSpectrum members used by power()/cgauss<2>).
struct GenConst {
    int N, half;
    int kmax;          // int(double(N/2)/k_cutoff + .5)           zeldovich.cpp:350
    int corner_modes;  //                                          zeldovich.cpp:353
    int qonemode, one_mode[3];
    int ablate;        // debug/tuning only (ZD_ABLATE env): skip parts of the generator; 0 in production
    double fundamental, fundamental2, k2_cutoff;
    int k2i_cut;       // smallest integer |k|^2 with double(|k|^2)*fundamental2 >= k2_cutoff (the rule of :353 on integers)
    // PowerSpectrum
    int pk_n, fixed_power, is_powerlaw;
    const double *pk_x, *pk_y, *pk_y2;
    const double2 *pk_tab;    // {P(k), 1/k^2} indexed by integer kx^2+ky^2+kz^2 (NULL: evaluate per mode)
    // k_genf: LDS image {directions, ln bins, 2^(j/64), spline segment records, segment LUT} (GenfTab, zd_kernels.hip)
    const double *genf_tab;
    int genf_n, genf_nseg;    // doubles in the image; spline segments (0 for a power law)
    double glut_x0, glut_inv_dx;
    const int *pk_lut;        // PK_LUT uniform cells over [x0, x_last] -> start segment
    double lut_x0, lut_inv_dx;
    double pk_norm, pk_smooth2, powerlaw_index;
    // PLT
    int qPLT, qPLTrescale;
    double f_cluster, target_f, ln_growth_ratio;  // log(a_NL/a0)
    const double *eig;
    long long eig_ppd;
    // k_genf, interpolated lookups: the table interpolated in (x, y) for every column of the slab being generated,
    // eig_lines[(cz * eig_rows + row of the slab) * N + x] = {e_x, e_y, e_z, lambda} at table cell cz along z (k_eig_lines);
    // a mode then blends two entries instead of eight table corners.  NULL: the direct lookup
    const double *eig_lines;
    int eig_rows;
    // local primordial non-Gaussianity (zeldovich.cpp:377-400): M(k) = 2 g c^2 T(k) k^2 / (3 Omega_M H0^2)
    int gen_phi;              // 1: emit phi = D / M (first f_NL pass)
    const zdfft::cplx *phik;  // non-NULL: D = phik[ky][z][x] * M (second pass; the zero rule is bypassed)
    const double *fnl_M;      // M by integer kx^2+ky^2+kz^2
    double fnl_pre, fnl_den, primordial_norm, n_s;
    // sum |D|^2 (Parseval form of output.cpp:197 density_variance) for the packed stores: NSLOT replicated slots
    double *var_slots;
    int accum_var;
    // RNG: state at the start of each ky plane (== reference v2rng[ky], power_spectrum.cpp:30-36)
    const zdpcg::u128 *row_state;
    // ZD_Version = 1 (zd_kernels_v1.hip): the accepted (phase1, phase2) pairs of cgauss<1> for the rows of the slab being
    // generated, [row of the slab][z][x]; NULL for the version-2 counter streams
    const double2 *v1dev;
};

// ZD_Version = 1: one mt19937 stream per yres (src/power_spectrum.cpp:18-25) between two launches of k_v1_draw: the state
// words at a regeneration boundary and the accepted pairs already drawn past the last mode served
constexpr int V1_QCAP = 1024;
struct V1Stream {
    uint32_t mt[624];
    uint32_t nq, pad[3];
    double2 q[V1_QCAP];
};

// Affine maps used by the generator's z-walk (set per launch geometry)
struct GenJumps {
    // The generator keeps its state ONE draw ahead of the mode's counter and consumes two draws per
    // mode, so each map advances (2*65536*drows - 1) draws; index 1 = the step crosses the z = N/2
    // wrap of the counter (rows N/2+1.. live at 65536-N+z, zeldovich.cpp:335).
    zdpcg::Affine fwd[2];   // z -> z + L         (next residue-fold term)
    zdpcg::Affine back[2];  // z -> z - (R-1)L + 1 (first term of the next k2; for R = 1: the next row)
    // the same moves for a mode whose draws are not needed (k_genf, all 64 modes of a wave zeroed):
    // 2*65536*drows draws in one map
    zdpcg::Affine fwd_full[2], back_full[2];
    // the mirrored walk of genf_tile_kz (PLT kinds: a thread also owns the lines L - k2, visited z' = N - z): every move negated —
    // z' -> z' - L inside a fold, z' -> z' + (R-1)L - 1 to the next line — with the same wrap variants and full-stride forms
    zdpcg::Affine mfwd[2], mback[2], mfwd_full[2], mback_full[2];
};

// genf_tile_kz (PLT kinds, kz mirror folded in): the lines k2 = 0 .. L/2 of the lower half in chunks of ZR; a lone last line
// (L/2 a multiple of ZR) is taken by the chunk before it.  0: no pairing (odd L)
ZD_HD int kz_chunks(int L, int ZR) {
    if (L % 2 || L < 4) return 0;
    const int lines = L / 2 + 1;
    return lines % ZR == 1 && lines > 1 ? lines / ZR : (lines + ZR - 1) / ZR;
}

// Addressing of the z-transformed block store ("BlockArray", include/block_array.h:26-35, re-laid
// out for the GPU).  Rows are grouped by the rank that generated them: rank g holds half-space rows
// g, g + G, g + 2G, ... at slots 0..Hq-1 and their Hermitian twins at slots Hq..2Hq-1 — the reference's
// "displaced twin" storage (zeldovich.cpp:453-466, block_array.cpp:487-491); the unused twin slot of
// ky = 0 is the Nyquist row ky = N/2, which is never read (treated as zero, zeldovich.cpp:644-650).
//
// Inside a chunk (= what one peer rank sends) the store is tiled in blocks of Bz planes x Bk row
// slots x N columns (~2 MB, one large page): the z stage writes ALL planes of a few rows and the y
// stage reads ALL rows of a few planes, so a plain [plane][row] or [row][plane] order makes one of
// them touch a different page (and the same HBM channel) with every access; with the tiling both
// touch N/8-ish pages per workgroup.  Element (chunk c, local plane zl, array a, slot s, column x):
//   c*chunk_stride + (s/Bk)*kb_stride + (zl/Bz)*zb_stride + a*a_stride + ((zl%Bz)*Bk + s%Bk)*N + x
struct StoreLayout {
    int N, half, Hq, narray;
    int lHq;         // log2(Hq): Hq, Zq, N are powers of two, so every division is a shift
    int lG;          // log2(number of ranks): half-space row ky belongs to rank ky mod G (cyclic: the rows near ky = 0
                     // carry most of the non-zero modes, contiguous blocks would leave rank 0 with up to 1.5x the work)
    int ky_stride;   // = G: rows ky0, ky0 + G, ... of a generator / z-FFT launch
    int lBk, lBz;    // log2 of the block edge in row slots / planes
    int rows_outer;  // order inside a block: 0 = [plane][slot][x], 1 = [slot][plane][x]
    int one_block;   // 1: single rank and Bk = 2*Hq, Bz = 1 -> slot s of (plane, array) is row s of one contiguous region
    int pitch;       // row pitch in elements (>= N)
    // all strides are in ROWS (32-bit: a 275 GB store has < 2^24 rows) so that one 32x32->64 multiply
    // per element turns a row index into an address
    int chunk_rows, kb_rows, zb_rows, a_rows;
    // pruning: a (kx,ky) column whose every kz mode is zeroed by the rule of zeldovich.cpp:350-353 is
    // identically zero after the z FFT; it is neither written by the z stage nor read by the y stage
    int prune, kmax;  // bits: 1 generator, 2 z FFT, 4 y FFT (per element), PRUNE_YTILE; bits 3..12 are tuning ablations
    double fund2, k2_cutoff;
    int nt;  // tuning (ZD_NT): non-temporal accesses, bit 0 y loads, 1 y stores, 2 x loads, 3 z stores, 4 z loads, 5 ring stores / 6 first-potential loads of k_yfft_f, 7 record stores
    // Plane-interleaved rows (round 5; the fused generator + z FFT of the packed PLT store, zd_kernels_fz.hip): 2^lq consecutive
    // planes share a row image, element (plane zl, column x) of a row sits at (x << lq) + (zl & (2^lq - 1)) of the row of plane
    // group zl >> lq; `pitch` is then the pitch of that interleaved row ((N + pad) << lq) and the strides count plane GROUPS.
    // A fused z workgroup holds ONE column's z lines (all planes, lanes along the planes): with lq = 2 its store instruction writes
    // 64-byte runs (4 planes x 16 B) instead of 16-byte pieces, the y stage's 8-line tile is 2 columns x 4 planes = one 128-byte
    // line, the x stage transforms the lines of a plane pair side by side.  0 everywhere else.
    int lq;
};

// StoreLayout::prune bit: the y stage skips whole column tiles without a live row.  Set only together with
// EpiConst::xdead_lo / hi, i.e. when the consumer of the y stage's output takes zeros for those columns (k_xfft*); the f_NL phi
// round (k_xphi / k_yfwd / k_zfwd read every column) keeps the tiles and the per-element rule only
constexpr int PRUNE_YTILE = 1 << 15;

// true iff every mode of column (kx, ky) (signed wavenumbers) is zero for all kz
ZD_HD bool column_is_zero(const StoreLayout &L, int kx, int ky) {
    if (!L.prune) return false;
    const int ax = kx < 0 ? -kx : kx, ay = ky < 0 ? -ky : ky;
    if (ax == L.kmax || ay == L.kmax) return true;
    // (kx^2+ky^2+kz^2)*fund2 >= (kx^2+ky^2)*fund2 in double arithmetic, so this implies the mode rule
    return L.k2_cutoff > 0 && (double) (kx * kx + ky * ky) * L.fund2 >= L.k2_cutoff;
}

ZD_HD void row_slot(const StoreLayout &L, int ky, int &chunk, int &slot) {
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
    chunk = kyh & ((1 << L.lG) - 1);
    slot  = (kyh >> L.lG) + (tw << L.lHq);
}
// row index of (chunk, local plane zl, array a, row slot); element offset = row * pitch + x
ZD_HD int store_row(const StoreLayout &L, int chunk, int zl, int a, int slot) {
    const int Bk = 1 << L.lBk, Bz = 1 << L.lBz;
    return chunk * L.chunk_rows + (slot >> L.lBk) * L.kb_rows + (zl >> L.lBz) * L.zb_rows + a * L.a_rows
           + (L.rows_outer ? (((slot & (Bk - 1)) << L.lBz) + (zl & (Bz - 1)))
                           : (((zl & (Bz - 1)) << L.lBk) + (slot & (Bk - 1))));
}
ZD_HD long long store_offset(const StoreLayout &L, int chunk, int zl, int a, int slot) {
    return (long long) store_row(L, chunk, zl, a, slot) * L.pitch;
}
// element offset of (chunk, local plane zl, array a, row slot, column x), plane-interleaved rows included (StoreLayout::lq)
ZD_HD long long store_elem(const StoreLayout &L, int chunk, int zl, int a, int slot, int x) {
    return (long long) store_row(L, chunk, zl >> L.lq, a, slot) * L.pitch + (((long long) x) << L.lq) + (zl & ((1 << L.lq) - 1));
}
ZD_HD long long row_offset(const StoreLayout &L, int zl, int a, int ky) {
    int c, s;
    row_slot(L, ky, c, s);
    return store_offset(L, c, zl, a, s);
}

// work item of the fused generator + z FFT (zd_kernels_fz.hip): `n` consecutive columns x0 .. x0 + n - 1 (not across x = N/2 | N/2 + 1,
// where the draw counter of a row jumps) of the half-space row with local index kyl
struct FzItem {
    int kyl, x0, n, pad;
};

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
    JOB_DENS    = 7,  // D -> array 0 self; conj -> twin         (qdensity == 2)
    // packed stores (no density field, see PACK_*):
    JOB_XV_SELF = 8,  // (i - f) sx D = F_x + i f F_x            (qx + i vx)
    JOB_XV_TWIN = 9,  // conj FFT[(i + f) sx D]
    JOB_FX      = 10, // i sx D of one residue; the two residues of a pass are combined into qx_r0 + i qx_r1
    // field store (PACK_ZAFIELD): the two potentials of the ZA displacement, per z-residue of the pass
    JOB_E       = 11, // (fundamental / k^2) D          q_x = d/dx, q_y = d/dy of its transform
    JOB_Z       = 12, // kz (fundamental / k^2) D       q_z = i x its transform
    // PLT field store (PACK_PLTFIELD): the six coefficient sums of the displacement / velocity components
    JOB_PX = 13, JOB_PY = 14, JOB_PZ = 15,      // s_x D, s_y D, s_z D          (q_j = i x its transform)
    JOB_PFX = 16, JOB_PFY = 17, JOB_PFZ = 18    // f s_x D, f s_y D, f s_z D    (v_j = i x its transform)
};
// What the store holds.  The reference always transforms the density (Re of its array 0) although only
// ZD_qdensity writes it out and only its sum of squares is reported; without ZD_qdensity the density is not
// transformed here (sum delta^2 = N^3 sum |D(k)|^2 comes from the generator), which leaves 3 (ZA) / 6 (PLT) real
// fields = 1.5 / 3 packed complex arrays instead of 2 / 4:
//   PACK_PLT3    arrays  qx + i vx | qy + i qz | vy + i vz
//   PACK_ZAPAIR  one pass carries TWO z-residues r0, r1 (the y/x transforms act per plane, so fields of
//                different planes pack into one complex array):  (qy + i qz)_r0 | (qy + i qz)_r1 | qx_r0 + i qx_r1
//   PACK_ZAFIELD the ZA displacement is the gradient of ONE scalar: after the z transform, q_x = i kx E, q_y = i ky E and
//                q_z = i Z with E = sum_kz (fund/k^2) D e^{..}, Z = sum_kz kz (fund/k^2) D e^{..}.  E is Hermitian and Z
//                anti-Hermitian in (kx, ky), so the store keeps E and Z of the two residues of a pass for the half-space
//                rows ky < N/2 ONLY (no Hermitian twins: 4 half fields = 2 arrays' worth instead of 3), and skips the
//                (kx, ky) columns the zero rule kills (FieldLayout below: 21.5 % at k_cutoff = 1).  That is what lets
//                PPD = 4096 run in 4 passes instead of 8 on one 288 GB GPU and halves the all-to-all volume between
//                GPUs.  The y pass builds, plane by plane, the three arrays (qx + i qy)_r0 | (qx + i qy)_r1 | qz_r0 + i qz_r1
//                (each potential feeds exactly one of them) into a small ring that the x pass consumes.
//   PACK_PLTFIELD the same for PLT: its six real fields are independent, so the store keeps the six sums X, Y, Z, fX, fY, fZ
//                (s_j D and f s_j D summed over kz; each anti-Hermitian in (kx, ky)) for the half-space rows — the size of
//                PACK_PLT3 minus the zero columns, but no mirrored twin stores in the z stage, 8-line z-FFT workgroups that
//                fit beside the generator, and whole-line reads in the y stage, which builds qx + i vx = i X - fX,
//                qy + i qz = i Y - Z, vy + i vz = i fY - fZ.
enum { PACK_NONE = 0, PACK_ZAPAIR = 1, PACK_PLT3 = 2, PACK_ZAFIELD = 3, PACK_PLTFIELD = 4 };
ZD_HD bool pack_is_fields(int pack) { return pack == PACK_ZAFIELD || pack == PACK_PLTFIELD; }

// Field store addressing.  chunk c = the rank that generated the rows (ky = c + G*slot), inside a chunk
// [plane zl][field f < nfield][row block][compact x][row in block].  A block keeps the columns x < split and x >= split + gap
// (kx in (-w, w) rounded out to FIELD_CW columns; the same table for every chunk, taken from the longest row of the
// block, ky = G * 8 * block), at position x (x < split) or x - gap.
constexpr int FIELD_CW = 32;  // compaction granularity in columns: a multiple of every z / y tile width
constexpr int FIELD_RB = 8;   // rows per block: element (slot, x) sits at base(slot / 8) + pos(x) * 8 + slot % 8, i.e. one
                              // column x 8 rows = one 128-byte line.  The y stage reads rows y > N/2 at column N - x: with
                              // plain rows that mirrored run of a 4-column tile starts 16 B before a line boundary (two
                              // lines fetched for one); with the blocks both the direct and the mirrored run of a tile
                              // are whole lines (measured: y stage 0.88 -> 0.7x s at PPD = 4096)
struct FieldRow {  // one record per BLOCK of FIELD_RB rows, 8 bytes: one load per row in the y stage
    int base;              // element offset of the block inside a (plane, field) image
    unsigned short split;  // first column not stored on the low side (N if the whole row is stored)
    unsigned short gap;    // columns skipped between the low and the high part
};
struct FieldLayout {
    int lG, lZq;                 // log2(ranks), log2(planes per chunk)
    int Zq;                      // planes per chunk (not a power of two for PPD = 2^a 3^b on several ranks)
    int nfield;                  // 4 (ZA: E0, Z0, E1, Z1), 6 (PLT: X, Y, Z, fX, fY, fZ), or 6 = ZA + the density sums D0, D1 (ndens = 2)
    int ndens;                   // 2: fields 4, 5 are the density sums of the two residues (ZD_qdensity = 1 on the composite grids)
    long long chunk_elems;       // elements per chunk = Zq * nfield * field_elems
    long long field_elems;       // elements per (plane, field) image = sum of the row lengths
    const FieldRow *rows;        // [Hq / FIELD_RB] device table
};
struct JobList {
    int n;
    int pack;
    int kind[8];
    int arr[8];   // destination array of the store
    int twin[8];  // 1: output goes (conjugated) to row N-ky, column N-kx only
    int res[8];   // PACK_ZAPAIR: which residue of the pass (0/1) the job's fold uses
};

struct EpiConst {
    int N, narray, icformat, recsize;
    int pack;      // PACK_*: field order of the store
    int z_pair;    // PACK_ZAPAIR: z distance between the two planes a store plane carries (R/2)
    int qPLT, qdensity;
    double vnorm;  // (sqrt(1+24 f_cluster)-1)/4 without PLT, 1 with   output.cpp:78-82
    // Field-store ring: the columns xdead_lo <= x <= xdead_hi are identically zero for every row (no (kx, ky) column with that kx
    // survives the zero rule of zeldovich.cpp:350-353: 1 column at k_cutoff = 1, half of them at k_cutoff = 2).  The y stage does
    // not transform or write tiles that lie wholly inside, the x stage takes zeros instead of reading them.  lo > hi: none
    int xdead_lo, xdead_hi;
};
ZD_HD bool x_is_dead(const EpiConst &ec, int x) { return x >= ec.xdead_lo && x <= ec.xdead_hi; }

// any even PPD (zd_kernels_any.hip): a length-n transform as a convolution of length M = 2^m >= 2n - 1 (Bluestein)
struct AnyTab {
    int n, M;
    double invM;
    const zdfft::cplx *twM;    // exp(2 pi i k / M), k < M
    const zdfft::cplx *chirp;  // c_m = exp(i pi m^2 / n), m < n
    const zdfft::cplx *fb;     // forward DFT_M of b_m = conj(c_m), |m| < n, wrapped
};
struct AnyLayout {  // store [plane][array][row y][x]
    int N, pitch, narray;
};

// device-side reductions (output.cpp:28-30,190-197): NSLOT replicated accumulators
constexpr int NSLOT = 64;
struct Reduce {
    double sumsq[NSLOT];
    // max_disp (output.cpp:190-193: largest |v| per axis, first occurrence in (z, y, x) order on a tie): bit pattern of |v|, and
    // (linear lattice index << 1) | (v < 0) of the record that holds it; updated as a pair under lock[slot] (zd_epi.h max_commit)
    unsigned long long maxabs[3][NSLOT];
    unsigned long long maxkey[3][NSLOT];
    unsigned int lock[NSLOT];
};

// y stage of the field stores: workgroup index -> (column tile of W columns, array a of 3).  Workgroups go to the 8 XCDs round-robin
// by their linear index and every XCD has its own L2, so the workgroups that read or write the same lines are made neighbours ON
// ONE XCD: the TPL tiles of a 128-byte ring line (W < 8), their mirror images (rows y > N/2 are read at column N - x: a tile's
// mirrored reads are its mirror tile's direct reads, off by one column), for each of the three arrays.
template <int NT, int W>
__device__ __forceinline__ void ytile_of(int id, int &tile, int &a) {
    constexpr int TPL = W >= 8 ? 1 : 8 / W, GS = 6 * TPL;
    if constexpr (NT % (16 * TPL) == 0) {
        const int xcd = id & 7, s = id >> 3;  // s: position in this XCD's stream
        // (groups interleaved over the XCDs; giving each XCD a contiguous range of groups measured 12 % slower)
        const int g = (s / GS) * 8 + xcd, m = s % GS;  // group of 2*TPL tiles, member
        // Order inside a group: the direct tiles (three arrays x the TPL tiles of a line, the two halves of a ring line next to
        // each other) first, their mirror tiles 3*TPL positions later.  A tile and its mirror tile read the same potential
        // lines; a workgroup starts every ~1 us on an XCD, and the two do best a few microseconds apart — close enough for the
        // L2 to still hold what the first one fetched, not so close that both ask for a line while it is on its way (measured:
        // asking at the same moment is the worst case).  y stage at PPD = 4096, distance in positions: 1 (tile, mirror, tile,
        // mirror) 733 ms, 2 (round 2's order) 713, 6 (this) 683-688, 12 693, 24 720, 48 742.  FETCH_SIZE is the same 1.4 TB per
        // step at distance 2 and 6 (profiles/r03a, r03b): the bytes fetched did not change, the time the requests take did.
        // (A tile reads its direct rows ascending, then its mirrored rows = its partner's direct rows descending; the partner runs
        // the same code, i.e. asks for the shared lines in exactly the reverse order — last fetched, first re-read.  Giving one
        // side the other order costs 17 %.)
        const int side = m / (3 * TPL), mm = m % (3 * TPL);
        a    = mm / TPL;
        tile = side ? NT - 1 - (TPL * g + mm % TPL) : TPL * g + mm % TPL;
    } else {
        tile = id % NT;
        a    = id / NT;
    }
}

#if defined(__HIPCC__)
// zero rule of LoadPlane (src/zeldovich.cpp:350-356)
__device__ __forceinline__ bool mode_is_zero(const GenConst &g, int kx, int ky, int kz, double k2) {
    const int ax = kx < 0 ? -kx : kx, ay = ky < 0 ? -ky : ky, az = kz < 0 ? -kz : kz;
    if (ax == g.kmax || az == g.kmax || ay == g.kmax) return true;
    if (!g.corner_modes && k2 >= g.k2_cutoff) return true;
    if (g.qonemode && !(kx == g.one_mode[0] && ky == g.one_mode[1] && kz == g.one_mode[2])) return true;
    return false;
}
#endif

}  // namespace zd
