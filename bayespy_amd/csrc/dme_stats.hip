// dme_stats.hip — statistics build for the Dirichlet-multinomial MLE on MI355X (gfx950).
//
// Replaces, for a whole count matrix at once, the reference's per-row
//   CompressedRowData.appendRow       DirichletEstimation/dirichletMultinomialEstimation.py:113-123
//   the CLI's mean accumulator        DirichletEstimation/findDirichletPrior.py:87-89
// by one streaming pass that builds VALUE histograms
//   hist[k][v] = #{i : x_ik == v},   histV[t] = #{i : sum_k x_ik == t}
// (dme_scan.hip turns them into U[k][n] = #{i : x_ik > n}, V[n] = #{i : n_i > n}).
//
// Kernel shape (HBM-bound; algorithmic bytes = N*K*sizeof(elem), read exactly once):
//   * one 64-lane wave owns whole rows: a lane loads up to J vectors of E elements (16 B when the
//     row pitch allows) of one row, so the row total is a DPP wave reduction and X is read once;
//     when a row is shorter than a wave (C = K/E < 64) a wave carries R = 64/C rows side by side;
//   * a lane keeps the same columns for every row, so the fp64 mean accumulators live in VGPRs and
//     are combined in a FIXED order (waves in LDS, workgroups in a second kernel): the result is
//     bit-reproducible run to run;
//   * counts 1..B go to an LDS-privatised histogram of packed 16-bit counters (two columns per
//     32-bit word, ds_add_u32, a workgroup handles <= 65535 rows so a counter cannot overflow);
//     zero cells hit no real bin: a whole zero dword is skipped, a zero half of a non-zero dword is
//     added to a sink row that is never flushed, so the hot loop has no per-element branch;
//   * counts > B (rare) and row totals >= TB go straight to 64-bit global atomics;
//   * at the end the workgroup dumps its LDS histogram image as it stands (coalesced 16-byte
//     stores); hist_reduce_kernel adds the images of all workgroups into hist / histV.
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "dme_common.h"

namespace {

struct StatsArgs {
    const void* X;
    long long N, K, ld;
    u64* hist;
    long long nbins;
    u64* histV;
    long long tbins;
    double* part;    // [gridDim.x][K] per-workgroup mean accumulators
    u32* hist_part;  // [gridDim.x][part_words] raw LDS histogram images, one per workgroup
    long long part_words;
    u64* status;
    int rows_per_wg;
    int C;    // vectors (chunks) per row = K / E
    int R;    // rows carried side by side by one wave (1 unless MULTI)
    int B;    // LDS-privatised count bins 1..B (row 0 of the LDS histogram is the sink)
    int BRW;  // 32-bit words per LDS bin row
    int TB;   // LDS-privatised row-total bins [0, TB)
    int TBpad;  // words from one copy of the row-total bins to the next (>= TB; odd when VC > 1, so
                // that equal totals of different row slots fall into different LDS banks)
    int VC;     // copies of the row-total bins (side-by-side rows: a wave's 64 / C rows add at once)
    int Vw;     // words of all copies together, a multiple of 4
    int WT;     // LDS table of row weights 1/(1+n) for n in [0, WT), 2 words each, at word Vw
    int hoff;   // word offset of the count histogram in LDS = Vw + 2*WT
    int CW;     // side-by-side rows: LDS words per (j, p) group of a bin row (C, or 64 = one per lane)
    int steps;  // > 0: interleaved row mapping, loop steps per workgroup; 0: one contiguous slab each
    int adjacent;  // a wave's RB rows of a loop step are adjacent rows (see the kernel)
    int RC;        // side-by-side rows: copies of a column's words in a bin row (1 ... R; row slot r
                   // adds to copy r % RC)
    int CL;        // side-by-side rows: lanes per row = ceil(C / J); lane c of a row holds the vectors
                   // c, c + CL, ... of it
    int tail;      // K % E when the rows end inside a vector (then K >= E), else 0
    u64* overflow;  // store for cells with v >= nbins (dme_common.h), or NULL: such cells are reported in status[1]
};

// Compile-time ablation switches for timing experiments (never set in the shipped build):
// -DDME_ABLATE=<bits>: 1 no LDS count atomics, 2 no mean accumulators, 4 no large-count path,
// 8 no row-total histogram.
#ifndef DME_ABLATE
#define DME_ABLATE 0
#endif
#define DBG(bit) ((DME_ABLATE) & (bit))

// row groups per register buffer of the one-vector-per-lane kernels (rows up to 1 KB)
#ifndef DME_RB1
#define DME_RB1 4
#endif
// dwords of a vector load handled under one exec mask in the uint16 cell loop (1, 2 or 4)
#ifndef DME_GROUP
#define DME_GROUP 1
#endif

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// 16-byte vectors are loaded from addresses that need only be element-aligned: global memory takes
// unaligned vector loads (one global_load_dwordx4 either way), so rows of any pitch and any K run as
// full vectors; an aligned address costs nothing extra.
typedef u32x4 u32x4_u __attribute__((aligned(2)));

template <int BYTES, int ND>
__device__ __forceinline__ void load_chunk(const void* p, u32 (&d)[ND]) {
    if constexpr (BYTES == 32) {
        u32x4 v = __builtin_nontemporal_load((const u32x4_u*)p);
        u32x4 u = __builtin_nontemporal_load((const u32x4_u*)p + 1);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        d[4] = u.x; d[5] = u.y; d[6] = u.z; d[7] = u.w;
    } else if constexpr (BYTES == 16) {
        u32x4 v = __builtin_nontemporal_load((const u32x4_u*)p);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if constexpr (BYTES == 8) {
        u32x2 v = __builtin_nontemporal_load((const u32x2*)p);
        d[0] = v.x; d[1] = v.y;
    } else if constexpr (BYTES == 4) {
        d[0] = __builtin_nontemporal_load((const u32*)p);
    } else {
        d[0] = (u32)__builtin_nontemporal_load((const unsigned short*)p);
    }
}

// Returns x through an empty asm: address arithmetic of the rare global-atomic paths that depends
// on it cannot be hoisted out of those paths into (spilled) loop-invariant registers.
__device__ __forceinline__ int opaque(int x) {
#ifndef DME_NO_OPAQUE
    asm volatile("" : "+v"(x));
#endif
    return x;
}

__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) {
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(u32, __builtin_elementwise_min(x, y));
}

// acc + x*w for the CLI's mean accumulator (DE/fdp.py:89 `priors[i] += data[i] * weightForMean`).
// One fused multiply-add: the product is not rounded before the add (the reference rounds it), a
// difference below the summation-order difference that already separates the two.
__device__ __forceinline__ double mean_add(double acc, double x, double w) {
#ifdef DME_NO_FMA
    return __dadd_rn(acc, __dmul_rn(x, w));
#else
    return __fma_rn(x, w, acc);
#endif
}

__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(u32, __builtin_elementwise_max(x, y));
}

// Threads per workgroup (one workgroup per CU) and row groups per register buffer, for J vectors of
// E elements of ES bytes per lane; the kernel and the host plan both take them from here.
//   threads: 1024 leaves 128 VGPRs per lane, which the 16-column uint16 kernel fits; wider lanes (and
//     int32's 16 columns) take 512 or 256.  uint16 rows of up to 8 vectors per lane fit 512 threads
//     (236 VGPRs at J = 8): eight waves per CU instead of four took K = 4096 from 3.28 to 5.79 TB/s;
//     the packed int32 kernel of that width spills there and keeps 256.
//   row groups: enough 16-byte loads in flight per CU (ES == 4 with E == 8 is the packed-int32 kernel:
//     its raw buffers are twice the size).
//   teams (TW > 1 waves share a row, see the kernel): 512 threads, one row group per buffer.
//   side-by-side rows of 3 / 4 / 5 vectors per lane: 1024 / 768 / 512 threads, one row group per buffer
//     (K = 200: 4.44 TB/s at 512 threads and two groups, 5.10 so; K = 600: 4.65 -> 5.09; 1024 threads
//     spill at 4, 768 at 5).  The packed int32 kernel spills 40-56 bytes at 1024 threads and is still
//     faster there than at 768 without (K = 2048: 5.51 against 4.72 TB/s; 512 threads with two row
//     groups per buffer: K = 1000 4.05 against 5.55).
//   uint16 rows of 3 / 4 / 5 vectors per lane likewise: 1024 / 768 / 768 threads (K = 1536: 4.96 -> 5.91
//     TB/s, K = 2048: 5.26 -> 5.97, K = 2560: 4.88 -> 6.10; 6 vectors spill at 768 threads: 5.02 -> 4.09)
constexpr int stats_block(int J, int E, int ES, int TW = 1, bool multi = false) {
    return (multi && J == 3) ? 1024 : (multi && J == 4) ? 768 : (multi && J == 5) ? 512
           : (multi && ES == 4 && E == 8 && J == 2) ? 768
           : (TW == 1 && ES == 2 && E == 8 && J == 3) ? 1024 : (TW == 1 && ES == 2 && E == 8 && J == 4) ? 768
           : (TW == 1 && ES == 2 && E == 8 && J == 5) ? 768
           : (TW > 1 && ES == 2 && J == 3) ? 1024 : (TW > 1 && ES == 2 && J == 4 && TW <= 4) ? 768
           : TW > 1 ? ((J * E > 16) ? 512 : 1024)
           : (J * E > 64 || (ES == 4 && J * E > 32)) ? 256 : ((J * E > 16) ? 512 : 1024);
}
constexpr int stats_rb(int J, int E, int ES, int TW = 1, bool multi = false) {
    return (multi && J >= 3) ? 1
           : (TW == 1 && ES == 2 && E == 8 && (J == 3 || J == 4)) ? 1
           : (TW == 1 && ES == 4 && E == 8 && J == 1) ? 2
           : (TW > 1 || (ES == 4 && E == 8)) ? 1
           : (J == 1)          ? DME_RB1
           : (J == 2)          ? 2
           : ((J == 3 || J == 4) && stats_block(J, E, ES) == 512) ? 2 : 1;
}
template <int J, int E, int ES, int TW = 1, bool MULTI = false>
struct StatsCfg {
    static constexpr int BLOCK = stats_block(J, E, ES, TW, MULTI);
    static constexpr int RB = stats_rb(J, E, ES, TW, MULTI);
};

// T: element type (unsigned short | int).  E: elements per vector (chunk).  J: vectors per lane
// and row.  MULTI: a wave carries R rows at once, each on CL = ceil(C / J) of its lanes.
// PACK (T = int, E = 8): int32 rows whose counts are, with rare exceptions, below 65536: a chunk
// is loaded as two 16-byte vectors, the exceptions (and negatives) are settled one by one, and the
// eight counts are packed into four dwords that then take the uint16 code - the int32 kernel pays
// per element what the uint16 kernel pays (~half the instructions of the plain int32 path).
// TW: waves that share a row (a team).  Rows of more than 8 vectors per lane (4 for the packed
// int32 kernel, whose fp64 accumulators and raw buffers fill the register file) are split over TW
// waves: wave t of a team takes the vectors 64 * (t + TW * j) + lane, the waves' partial row totals
// meet in LDS (a counter per team, two sets of slots), everything else is per wave as before.  The LDS image keeps its layout (vector index t + TW * j where a one-wave row has j).
// SCAN (side-by-side rows of more than 8 lanes): row totals from a DPP prefix sum over the wave.
// WIDE (side-by-side uint16 rows): 32-bit LDS counters, one column per word, so a workgroup is not
// held to 65535 rows: narrow matrices of very many rows then leave one image per CU, not one per
// 60000 rows (K = 8, 500 M rows: 8448 images of 100 KB to write and to sum).
// SUB (with WIDE; K = 2 or 4 in rows that lie back to back): a 16-byte vector holds SUB whole rows,
// a lane settles them by itself (row totals inside the lane).
// TAIL: the rows end inside their last vector (a.tail elements of it belong to the row).
// EARLY (PACK): a buffer's counts are packed to 16-bit pairs as soon as they have arrived and the buffer is
// refilled at once, two steps ahead: two buffers of loads are in flight while the cells are counted, for
// eight more registers.  What the widest rows and every shape that leaves lanes idle were short of
// (int32 K = 16 384: 3.6 -> 5.8 TB/s, K = 10: 3.4 -> 4.4, K = 600: 3.8 -> 4.7); shapes that fill their waves
// lose 4-7 % to it and run without (the plan decides).
template <typename T, int E, int J, bool MULTI, bool PACK, int TW, bool SCAN, bool WIDE = false, int SUB = 1,
          bool TAIL = false, bool EARLY = false>
__global__ __launch_bounds__((StatsCfg<J, E, (int)sizeof(T), TW, MULTI>::BLOCK)) void stats_dense_kernel(StatsArgs a) {
    static_assert(TW == 1 || !MULTI, "teams are for long rows");
    static_assert(!WIDE || (MULTI && (sizeof(T) == 2 || PACK) && E == 8), "32-bit counters: side-by-side rows of 16-bit cells");
    static_assert(SUB == 1 || J == 1, "rows inside a vector: one vector per lane");
    static_assert(!TAIL || (E * sizeof(T) >= 16 && SUB == 1), "row ends inside full vectors only");
    static_assert(SUB == 1 || (WIDE && !SCAN && (SUB == 2 || SUB == 4)), "rows inside a vector need WIDE");
    static_assert(!EARLY || PACK || sizeof(T) == 2, "early refill: kernels that count 16-bit pairs");
    constexpr int KC = E / SUB;  // columns of a row inside a vector (SUB > 1)
    constexpr bool RAW16 = sizeof(T) == 2;  // the buffers hold packed 16-bit counts
    constexpr bool IS16 = RAW16 || PACK;    // the cells are processed as packed 16-bit counts
    constexpr int BYTES = E * (int)sizeof(T);
    constexpr int ND = BYTES >= 4 ? BYTES / 4 : 1;  // dwords held per vector
    constexpr int NDP = IS16 ? (E >= 2 ? E / 2 : 1) : ND;  // dwords per vector once packed
    constexpr int PW = WIDE ? KC : (E >= 2 ? E / 2 : 1);  // LDS words per vector (2 columns per word, or 1)
    constexpr int BLOCK = StatsCfg<J, E, (int)sizeof(T), TW, MULTI>::BLOCK;
    constexpr int W = BLOCK / 64;
    constexpr int WR = W / TW;  // rows a workgroup has in hand per register-buffer slot
    constexpr int RB = StatsCfg<J, E, (int)sizeof(T), TW, MULTI>::RB;
    typedef typename std::conditional<RAW16, u32, u64>::type SumT;

    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw = TW > 1 ? wave % TW : 0;    // this wave's place in its team
    const int wrow = TW > 1 ? wave / TW : wave;  // the team's (= the row's) place in the workgroup
    const int C = a.C;
    const int R = MULTI ? a.R : 1;
    const int CL = MULTI ? a.CL : 64;  // lanes per row
    const int r = MULTI ? lane / CL : 0;
    const int c = MULTI ? lane - r * CL : lane;
    const bool lane_has_row = MULTI ? (r < R) : true;
    const int CC = MULTI ? CL : 64;  // column chunks a wave covers per vector index j
    const int CW = MULTI ? a.CW : 64;  // LDS words per (j, p) group of a bin row: C, or one per lane
    const int B = a.B, BRW = a.BRW, TB = a.TB;
    u32* ldsV = lds;
    u32* ldsH = lds + a.hoff;
    const double* ldsW = (const double*)(lds + a.Vw);
    const int WT = a.WT;
    const u32 vcopy = MULTI ? (u32)((r & (a.VC - 1)) * a.TBpad) : 0u;  // this row slot's copy of the bins
    const int nwords = (int)a.part_words;  // TB + (B+1)*BRW rounded up to a multiple of 4
    for (int i = tid; i < nwords; i += BLOCK) lds[i] = 0;
    // Data-error counters and the maxima seen by the rare global-atomic paths live in LDS, not in
    // registers (the maxima of the LDS-privatised bins are taken by hist_reduce_kernel):
    // [1] cells >= nbins, [2] rows >= tbins, [3] negative cells, [4] max count, [5] max row total,
    // [7] team waits that timed out (an internal fault: the result is incomplete).
    __shared__ u64 sh_stat[8];
    __shared__ u64 sh_tot[2][TW > 1 ? W : 1][RB];  // teams: the waves' partial row totals
    __shared__ u32 sh_team[TW > 1 ? WR : 1];        // teams: parts published so far
    // Packed int32 kernels (at their 128-register limit): the overflow store's address waits in LDS for
    // the rare branch that needs it.  Read from the kernel argument there, it stayed live across the
    // main loop (K = 8192: 1.85 -> 1.92 ms for 8 GB; from LDS 1.62 ms).  The uint16 kernels measured
    // SLOWER that way (K = 16 384: 1.55 -> 1.71 ms) and read the argument.
    __shared__ u64* sh_ovf;
    if constexpr (PACK) {
        if (tid == 0) sh_ovf = a.overflow;
    }
    if (tid < 8) sh_stat[tid] = 0;
    if (tid < (TW > 1 ? WR : 1)) sh_team[tid] = 0;
    __syncthreads();
    // weightForMean = 1.0 / (1.0 + sumData) (DE/fdp.py:88), tabulated once per workgroup
    for (int n = tid; n < WT; n += BLOCK)
        ((double*)(lds + a.Vw))[n] = __ddiv_rn(1.0, __dadd_rn(1.0, (double)n));
    __syncthreads();

    double acc[J][E];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) acc[j][e] = 0.0;
    u32 team_target = 0;
    const u32 Bpk = (u32)a.B | ((u32)a.B << 16);

    // Row mapping.  A loop step retires a tile of S = W * RB adjacent row groups; a wave takes RB
    // of them.
    //  * Interleaved tiles (a.steps > 0): workgroup b takes tiles b, b + gridDim.x, ..., so at any
    //    moment the whole chip reads one window of gridDim.x tiles (16 MB at K = 1000) that moves
    //    through the matrix, instead of gridDim.x slabs spread over all of it: the window covers a
    //    handful of pages and the kernel's time no longer depends on where the allocation happens
    //    to lie (3.32 - 3.57 ms on the same data in different buffers with one slab per
    //    workgroup).  Contiguous (a.steps == 0): workgroup b owns the rows_per_wg rows of slab b.
    //  * Within a tile wave w takes rows w, w + W, ... (a.adjacent == 0) or the RB adjacent rows
    //    w * RB ... (a.adjacent != 0).  When the row pitch is not a multiple of the 128-byte cache
    //    line, neighbouring rows share a line and the non-temporal stream fetches it once per
    //    load instruction that touches it, unless the two loads come from one wave back to back:
    //    adjacent rows in one wave halve those double fetches (K = 1000: 3.32 -> 3.25 ms).  With
    //    line-aligned rows there is nothing to save and the strided order measured 1-4 % faster
    //    (profiles/r1_notes.md).
    constexpr long long S = (long long)WR * RB;  // row groups a workgroup retires per loop step
    // (the packed int32 kernel sits at its 128-register limit and measured 1-2 % slower
    // interleaved: it keeps the slabs)
    const bool inter = !PACK && a.steps > 0;
    const bool adjacent = a.adjacent != 0;
    const long long row0 = inter ? 0 : (long long)blockIdx.x * a.rows_per_wg;
    const long long row1 = inter ? a.N : min(a.N, row0 + (long long)a.rows_per_wg);
    const long long nrows = row1 > row0 ? row1 - row0 : 0;
    // loop steps of this workgroup (rows beyond row1 load as zeros), row groups between its tiles,
    // its first tile, and the place of this wave's rb-th row group inside a tile
    const long long nsteps = inter ? (long long)a.steps : ((nrows + R - 1) / R + S - 1) / S;
    const long long tile_pitch = inter ? (long long)gridDim.x * S : S;
    const long long tile0 = inter ? (long long)blockIdx.x * S : 0;
    int goff[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) goff[rb] = adjacent ? wrow * RB + rb : rb * WR + wrow;
    const u32 binRowBytes = (u32)BRW * 4u;
    // byte offset of this lane's word 0 inside a bin row; word (j, p) is (j*PW + p)*CW words on.
    // Many side-by-side rows (a.CW == 64: 16 or more per wave) keep a word per LANE, not per column
    // chunk: the wave's rows would otherwise add equal counts of one column to one word, and
    // same-address LDS atomics serialise (K = 16: 1.25 -> 0.89 ms for 4 GB; hist_reduce_kernel adds
    // the row slots' words of a column together).  With few rows per wave the narrower layout
    // measured 1 % faster.
    // As many copies as the count bins leave room for (a.RC): counts beyond the LDS bins go to global
    // atomics, which for a handful of columns all land on a few addresses (K = 8 with counts up to
    // 500: 72 ms for 8 GB with a word per lane and 150 bins, against 2.3 ms).
    const u32 laneWordByte =
        (u32)(MULTI ? (r % a.RC) * CL + c : c) * 4u + (u32)(tw * PW * CW) * 4u;
    char* ldsHb = (char*)ldsH;
    const T* Xp = (const T*)a.X;
    const long long nbins = a.nbins, tbins = a.tbins;

    struct Buf {
        u32 d[RB][J][ND];
    };

    // issue the loads of this wave's RB row groups of the tile at row group `tile` (nothing is
    // waited for here; a tile beyond the last one holds no rows below row1)
    auto load_buf = [&](Buf& b, long long tile) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long long rowi = row0 + (tile + goff[rb]) * R + r;
            const bool ok = lane_has_row && (rowi < row1);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int chunk = MULTI ? c + CL * j : c + 64 * (tw + TW * j);
                if (ok && chunk < C) {
                    if (TAIL && __builtin_expect(chunk == C - 1 && rowi == a.N - 1, 0)) {
                        // The matrix ends inside this vector (the last vector of the last row; the
                        // last vectors of the other rows run on into the next row and are masked in
                        // process_buf): load the E elements that END with the row (K >= E: nothing
                        // beyond the matrix is touched) and move the a.tail elements that belong to
                        // this vector to its front.
                        u32 w[ND];
                        load_chunk<BYTES, ND>(Xp + rowi * a.ld + a.K - E, w);
                        const int sh = E - a.tail;  // elements to drop from the front: 1 .. E - 1
                        const int ds = sizeof(T) == 2 ? sh >> 1 : sh;  // whole dwords
                        u32 t[ND + 1];
#pragma unroll
                        for (int q = 0; q < ND; ++q) {
                            t[q] = 0;
#pragma unroll
                            for (int sft = 0; sft < ND; ++sft)
                                if (q + sft < ND && ds == sft) t[q] = w[q + sft];
                        }
                        t[ND] = 0;
                        const bool half = sizeof(T) == 2 && (sh & 1);
#pragma unroll
                        for (int q = 0; q < ND; ++q)
                            b.d[rb][j][q] = half ? (t[q] >> 16) | (t[q + 1] << 16) : t[q];
                    } else {
                        load_chunk<BYTES, ND>(Xp + rowi * a.ld + (long long)chunk * E, b.d[rb][j]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < ND; ++q) b.d[rb][j][q] = 0;
                }
            }
        }
    };

    // side-by-side rows: the total of this lane's row from the lanes' sums.  Rows of more than 8
    // lanes (SCAN): a prefix sum over the wave in DPP steps, then the difference of its values at
    // the row's last lane and at the lane before its first (two ds_bpermute; K = 200: 2.20 -> 2.01
    // ms for 8 GB).  Shorter rows: a shuffle scan inside the row (log2(C) + 1 ds_bpermute, which
    // measured faster up to C = 8; a kernel that holds both measured slower than either).
    const int lastLane = min(r * CL + CL - 1, 63), prevLane = max(r * CL - 1, 0);
    auto multi_total = [&](u32 s) -> u32 {
        if constexpr (SCAN) {
            const u32 incl = dme_wave_incl_scan_u32(s);
            const u32 hi = (u32)__shfl((int)incl, lastLane);
            const u32 lo = (u32)__shfl((int)incl, prevLane);
            return r > 0 ? hi - lo : hi;
        } else {
            u32 incl = s;
#pragma unroll
            for (int dd = 1; dd < 8 && dd < CL; dd <<= 1) {  // lanes of one row: c < CL
                u32 up = __shfl_up(incl, dd);
                if (c >= dd) incl += up;
            }
            return (u32)__shfl((int)incl, lastLane);
        }
    };

    // Cells with a count of nbins or more (rarer still than the rare global-atomic paths below) go to the
    // overflow store from ONE copy of the store's code per site: the unrolled per-element code only
    // marks them in a bit mask, and this loop - not unrolled, the value picked by a select chain -
    // walks the marks.  (With the store's code inlined per element the rare branches grew the team
    // kernels by a third and the widest rows lost 4-15 %.)  `half`: vals holds packed 16-bit pairs.
    auto overflow_cells = [&](u32 ovm, auto& vals, long long k0, bool half, int kmod) {
        u64* ovf = a.overflow;
        if constexpr (PACK) ovf = *(u64* volatile*)&sh_ovf;
#pragma nounroll
        for (u32 m = ovm; m; m &= m - 1) {
            const int e = __builtin_ctz(m);
            const int t = half ? e >> 1 : e;
            u32 v = 0;
#pragma unroll
            for (int i = 0; i < (int)(sizeof(vals) / sizeof(vals[0])); ++i) v = (t == i) ? vals[i] : v;
            if (half) v = (e & 1) ? v >> 16 : v & 0xffffu;
            const long long k = kmod ? (long long)(e % kmod) : k0 + e;
            if (!dme_overflow_put(ovf, (u64)k, (u64)v, 1ULL)) atomicAdd(&sh_stat[1], 1ULL);
        }
    };

    auto process_buf = [&](Buf& b, long long tile, int parity, long long reload_tile) {
        // rows that end inside a vector: clear what the last vector of a row holds beyond the row
        if constexpr (TAIL) {
            // what the last vector of a row keeps (a.tail elements)
            u32 tmask[ND];
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                if constexpr (sizeof(T) == 2)
                    tmask[q] = (2 * q < a.tail ? 0xffffu : 0u) | (2 * q + 1 < a.tail ? 0xffff0000u : 0u);
                else
                    tmask[q] = q < a.tail ? 0xffffffffu : 0u;
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int chunk = MULTI ? c + CL * j : c + 64 * (tw + TW * j);
                if (chunk == C - 1) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int q = 0; q < ND; ++q) b.d[rb][j][q] &= tmask[q];
                }
            }
        }
        // ---- row totals of all RB row groups first (independent DPP chains interleave) ----------
        SumT tot[RB];
        u32 tsub[RB][SUB];  // SUB > 1: the totals of the rows inside the lane's vector
        u32 mraw = 0;       // int32: the lane's largest raw count over the buffer (negatives read huge)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if constexpr (SUB > 1) {
                constexpr int DPS = ND / SUB;  // dwords of a row
#pragma unroll
                for (int sr = 0; sr < SUB; ++sr) {
                    u32 s = 0;
#pragma unroll
                    for (int q = 0; q < DPS; ++q) s = __builtin_amdgcn_sad_u16(b.d[rb][0][sr * DPS + q], 0u, s);
                    tsub[rb][sr] = s;
                }
                tot[rb] = 0;
            } else if constexpr (RAW16) {
                u32 s = 0;
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int q = 0; q < ND; ++q) s = __builtin_amdgcn_sad_u16(b.d[rb][j][q], 0u, s);
                if constexpr (MULTI) {
                    tot[rb] = multi_total(s);
                } else {
                    tot[rb] = dme_wave_sum_u32(s);
                }
            } else {
                // Unsigned maximum of the lane's counts (a negative count reads as a huge value).
                // While every count of the wave is below 2^20 the 32-bit sums cannot wrap
                // (64 lanes x 64 counts x 2^20 = 2^32 is never reached with J*E <= 64 ... < 2^32)
                // and the cheap reduction is exact; otherwise (rare) sum the positive parts in 64 bits.
                u32 m = 0, s32 = 0;
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int q = 0; q < ND; ++q) {
                        m = max(m, b.d[rb][j][q]);
                        s32 += b.d[rb][j][q];
                    }
                mraw = max(mraw, m);
                if (__builtin_expect(!__any(m >= (1u << 20)), 1)) {
                    if constexpr (MULTI) {
                        tot[rb] = (SumT)multi_total(s32);
                    } else {
                        tot[rb] = (SumT)dme_wave_sum_u32(s32);
                    }
                } else {
                    u64 s = 0;
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int q = 0; q < ND; ++q) {
                            const int v = (int)b.d[rb][j][q];
                            s += (u64)(v > 0 ? v : 0);
                        }
                    if constexpr (MULTI) {
                        u64 incl = s;
#pragma unroll
                        for (int dd = 1; dd < 64 && dd < CL; dd <<= 1) {  // lanes of one row: c < CL
                            u64 up = __shfl_up(incl, dd);
                            if (c >= dd) incl += up;
                        }
                        int last = r * CL + CL - 1;
                        tot[rb] = __shfl(incl, last > 63 ? 63 : last);
                    } else {
                        tot[rb] = dme_wave_sum_u64(s);
                    }
                }
            }
        }
        // counts packed to 16-bit pairs; those that do not fit (>= 65536, or negative) are marked, packed
        // as zeros, and read again from memory when their turn comes
        u32 P[EARLY ? RB : 1][EARLY ? J : 1][NDP];
        u32 rmask[EARLY ? RB : 1][EARLY ? J : 1];
        if constexpr (EARLY && !PACK) {
            // uint16 rows are 16-bit pairs as they stand: the copy costs registers, no instructions
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int q = 0; q < NDP; ++q) P[rb][j][q] = b.d[rb][j][q];
            load_buf(b, reload_tile);
        }
        if constexpr (EARLY && PACK) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    u32 rm = 0;
                    if (__builtin_expect(mraw > 0xFFFFu, 0)) {
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (b.d[rb][j][e] > 0xFFFFu) {
                                rm |= 1u << e;
                                b.d[rb][j][e] = 0;
                            }
                    }
                    rmask[rb][j] = rm;
#pragma unroll
                    for (int q = 0; q < 4; ++q) P[rb][j][q] = b.d[rb][j][2 * q] | (b.d[rb][j][2 * q + 1] << 16);
                }
            load_buf(b, reload_tile);
        }
        if constexpr (TW > 1) {
            // The team's waves add their parts.  No workgroup barrier: that would march all the
            // waves of the CU in step (every wave waiting for memory at once, then every wave at
            // the LDS at once; K = 4096 int32: 1.81 ms against 1.50 ms with free-running teams), so a team meets on
            // a counter of its own.  A wave's LDS operations execute in the order it issues them:
            // the slots are written before the counter is raised, and a mate that has seen the
            // count reads the slots after.  Two sets of slots: a wave can be one step ahead of its
            // mates and no more.  Every wave of a team runs the same nsteps steps, so each wait
            // ends.  (Relaxed atomics and compiler barriers only: a release / acquire pair here
            // would also wait for the global loads in flight.)
            // When the team is the whole workgroup (WR == 1) the barrier is the cheaper meeting
            // (K = 16384 int32: 1.93 ms against 3.15 ms with the counter).
            if constexpr (WR > 1) {
                if (lane == 0) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) sh_tot[parity][wave][rb] = (u64)tot[rb];
                    asm volatile("" ::: "memory");
                    __hip_atomic_fetch_add(&sh_team[wrow], 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                team_target += TW;
                // Bounded (MI355X_MICROARCH.md: bound every spin): a mate that never arrives - only
                // an edit that lets one wave of a team leave the loop early could cause it - ends the
                // wait after ~2^22 sleeps (seconds) and raises status[7]: the pass then reports
                // itself incomplete instead of hanging the CU.
                u32 spins = 0;
                while (__hip_atomic_load(&sh_team[wrow], __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP) < team_target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_expect(++spins > (1u << 22), 0)) {
                        if (lane == 0) atomicAdd(&sh_stat[7], 1ULL);
                        break;
                    }
                }
                asm volatile("" ::: "memory");
            } else {
                if (lane == 0) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) sh_tot[parity][wave][rb] = (u64)tot[rb];
                }
                __syncthreads();
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                u64 t = 0;
#pragma unroll
                for (int i = 0; i < TW; ++i) t += sh_tot[parity][wrow * TW + i][rb];
                tot[rb] = (SumT)t;
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long long pg = tile + goff[rb];
            if (row0 + pg * R >= row1) continue;  // wave-uniform: no such rows
            const SumT total = tot[rb];
            double w = 0.0;
            double wv[SUB];  // the weight of every row inside the vector (SUB > 1)
            if constexpr (SUB > 1) {
                const long long rowi = row0 + pg * R + r;
                const bool live = lane_has_row && rowi < row1;
#pragma unroll
                for (int sr = 0; sr < SUB; ++sr) {
                    const u32 ts = tsub[rb][sr];
                    if (live && ts > 0 && !DBG(8)) {
                        if (ts < (u32)TB) {
                            atomicAdd(&ldsV[vcopy + ts], 1u);
                        } else {
                            atomicMax(&sh_stat[5], (u64)ts);
                            if ((long long)ts < tbins) atomicAdd(&a.histV[ts], 1ULL);
                            else atomicAdd(&sh_stat[2], 1ULL);
                        }
                    }
                    wv[sr] = ts < (u32)WT ? ldsW[ts] : __ddiv_rn(1.0, __dadd_rn(1.0, (double)ts));
                }
            } else if constexpr (MULTI) {
                // ---- row-total histogram: the first lane of every row ----------------------
                const long long rowi = row0 + pg * R + r;
                if (lane_has_row && rowi < row1 && c == 0 && total > 0 && !DBG(8)) {
                    if ((u64)total < (u64)TB) {
                        // Rows of similar totals (coin-flip data: one fixed number of trials) would
                        // send the wave's adds to a few LDS words, which serialises them: the
                        // row slots of a wave are spread over VC copies of the bins.
                        atomicAdd(&ldsV[vcopy + (u32)total], 1u);
                    } else {
                        atomicMax(&sh_stat[5], (u64)total);
                        if ((long long)total < tbins) atomicAdd(&a.histV[total], 1ULL);
                        else atomicAdd(&sh_stat[2], 1ULL);
                    }
                }
                if ((u64)total < (u64)WT) w = ldsW[(u32)total];
                else w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
            } else {
                // ---- row-total histogram: `total` is wave-uniform, so these are scalar branches --
                if ((u64)total < (u64)TB) {
                    if (lane == 0 && tw == 0 && total > 0 && !DBG(8)) atomicAdd(&ldsV[(u32)total], 1u);
                } else if (!DBG(8)) {
                    atomicMax(&sh_stat[5], (u64)total);
                    if (lane == 0 && tw == 0) {
                        if ((long long)total < tbins) atomicAdd(&a.histV[total], 1ULL);
                        else atomicAdd(&sh_stat[2], 1ULL);
                    }
                }
                if ((u64)total < (u64)WT) w = ldsW[(u32)total];
                else w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
            }
            if constexpr (SUB == 1) wv[0] = w;
            // ---- cells --------------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int chunk = MULTI ? c + CL * j : c + 64 * (tw + TW * j);
                if constexpr (IS16) {
                    u32 dwl[NDP];
                    if constexpr (PACK) {
                        // counts >= 65536 and negatives (huge as unsigned) are settled here and
                        // cleared; what is left fits 16 bits
                        u32 mr = 0;
                        if constexpr (EARLY) {
                            const u32 rm = rmask[rb][j];
                            if (__builtin_expect(rm != 0, 0)) {
                                const long long rowi = row0 + pg * R + r;
#pragma nounroll
                                for (u32 mm = rm; mm; mm &= mm - 1) {
                                    const int e = __builtin_ctz(mm);
                                    const long long k = (long long)opaque(chunk) * E + e;
                                    const int v = (int)Xp[rowi * a.ld + k];
                                    if (v < 0) {
                                        atomicAdd(&sh_stat[3], 1ULL);
                                    } else {
                                        atomicMax(&sh_stat[4], (u64)v);
                                        if ((long long)v < nbins) {
                                            atomicAdd(&a.hist[k * nbins + v], 1ULL);
                                        } else {
                                            u64* ovf = *(u64* volatile*)&sh_ovf;
                                            if (!dme_overflow_put(ovf, (u64)k, (u64)v, 1ULL)) atomicAdd(&sh_stat[1], 1ULL);
                                        }
#pragma unroll
                                        for (int t = 0; t < E; ++t)
                                            acc[j][t] = (t == e) ? mean_add(acc[j][t], (double)v, w) : acc[j][t];
                                    }
                                }
                            }
                        } else {
                            mr = b.d[rb][j][0];
#pragma unroll
                            for (int e = 1; e < E; ++e) mr = max(mr, b.d[rb][j][e]);
                        }
                        if (__builtin_expect(mr > 0xFFFFu, 0)) {
                            u32 ovm = 0;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const int v = (int)b.d[rb][j][e];
                                if (v < 0) {
                                    atomicAdd(&sh_stat[3], 1ULL);
                                    b.d[rb][j][e] = 0;
                                } else if (v > 0xFFFF) {
                                    atomicMax(&sh_stat[4], (u64)v);
                                    acc[j][e] = mean_add(acc[j][e], (double)v, w);
                                    if ((long long)v < nbins) {
                                        const long long k = (long long)opaque(chunk) * E + e;
                                        atomicAdd(&a.hist[k * nbins + v], 1ULL);
                                        b.d[rb][j][e] = 0;
                                    } else {
                                        ovm |= 1u << e;
                                    }
                                }
                            }
                            if (__builtin_expect(ovm != 0, 0)) {
                                overflow_cells(ovm, b.d[rb][j], (long long)opaque(chunk) * E, false, 0);
#pragma unroll
                                for (int e = 0; e < E; ++e)
                                    if ((ovm >> e) & 1u) b.d[rb][j][e] = 0;
                            }
                        }
                        if constexpr (EARLY) {
#pragma unroll
                            for (int q = 0; q < NDP; ++q) dwl[q] = P[EARLY ? rb : 0][EARLY ? j : 0][q];
                        } else {
#pragma unroll
                            for (int q = 0; q < NDP; ++q)
                                dwl[q] = b.d[rb][j][2 * q] | (b.d[rb][j][2 * q + 1] << 16);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < NDP; ++q) dwl[q] = EARLY ? P[EARLY ? rb : 0][EARLY ? j : 0][q] : b.d[rb][j][q];
                    }
                    // rare path first: any count > B in this vector?  packed 16-bit max over the
                    // vector, then min(m, B|B) != m  <=>  some half exceeds B
                    u32 m = dwl[0];
#pragma unroll
                    for (int q = 1; q < NDP; ++q) m = pk_max_u16(m, dwl[q]);
                    if (__builtin_expect(pk_min_u16(m, Bpk) != m, 0) && !DBG(4)) {
                        u32 ovm = 0;
#pragma unroll
                        for (int q = 0; q < NDP; ++q)
#pragma unroll
                            for (int h = 0; h < (E >= 2 ? 2 : 1); ++h) {
                                const u32 v = h ? (dwl[q] >> 16) : (dwl[q] & 0xffffu);
                                if (v > (u32)B) {
                                    atomicMax(&sh_stat[4], (u64)v);
                                    acc[j][2 * q + h] = mean_add(acc[j][2 * q + h], (double)v, wv[(2 * q + h) / KC % SUB]);
                                    if ((long long)v < nbins) {
                                        const long long k = SUB > 1 ? (long long)((2 * q + h) % KC)
                                                                    : (long long)opaque(chunk) * E + 2 * q + h;
                                        atomicAdd(&a.hist[k * nbins + v], 1ULL);
                                        dwl[q] &= h ? 0x0000ffffu : 0xffff0000u;  // done: clear it
                                    } else {
                                        ovm |= 1u << (2 * q + h);
                                    }
                                }
                            }
                        if (__builtin_expect(ovm != 0, 0)) {
                            overflow_cells(ovm, dwl, SUB > 1 ? 0LL : (long long)opaque(chunk) * E, true,
                                           SUB > 1 ? KC : 0);
#pragma unroll
                            for (int q = 0; q < NDP; ++q)
                                dwl[q] &= (((ovm >> (2 * q)) & 1u) ? 0xffff0000u : 0xffffffffu) &
                                          (((ovm >> (2 * q + 1)) & 1u) ? 0x0000ffffu : 0xffffffffu);
                        }
                    }
                    // one exec-masked region per DME_GROUP dwords: a lane enters it when any of its
                    // 2*DME_GROUP counts is non-zero; zero halves inside go to the sink row
                    constexpr int QG = (NDP % DME_GROUP == 0) ? DME_GROUP : 1;
#pragma unroll
                    for (int q0 = 0; q0 < NDP; q0 += QG) {
                        u32 any = dwl[q0];
#pragma unroll
                        for (int t = 1; t < QG; ++t) any |= dwl[q0 + t];
                        if (__builtin_expect(any != 0, 1)) {  // some lane of the wave almost always is
#pragma unroll
                            for (int t = 0; t < QG; ++t) {
                                const int q = q0 + t;
                                const u32 dw = dwl[q];
                                const u32 lo = dw & 0xffffu;
                                if constexpr (WIDE) {
                                    // a word per column: columns 2q, 2q + 1 of the vector's row
                                    const u32 hi = dw >> 16;
                                    const u32 wb0 = laneWordByte + (u32)((j * PW + (2 * q) % KC) * CW) * 4u;
                                    const u32 wb1 = laneWordByte + (u32)((j * PW + (2 * q + 1) % KC) * CW) * 4u;
                                    if (!DBG(1)) {
                                        atomicAdd((u32*)(ldsHb + (__umul24(lo, binRowBytes) + wb0)), 1u);
                                        atomicAdd((u32*)(ldsHb + (__umul24(hi, binRowBytes) + wb1)), 1u);
                                    }
                                    if (!DBG(2)) {
                                        const double ww = wv[(2 * q) / KC % SUB];
                                        acc[j][2 * q] = mean_add(acc[j][2 * q], (double)lo, ww);
                                        acc[j][2 * q + 1] = mean_add(acc[j][2 * q + 1], (double)hi, ww);
                                    }
                                    continue;
                                }
                                const u32 wb = laneWordByte + (u32)((j * TW * PW + q) * CW) * 4u;
                                if (!DBG(1))
                                    atomicAdd((u32*)(ldsHb + (__umul24(lo, binRowBytes) + wb)), 1u);
                                if (!DBG(2)) acc[j][2 * q] = mean_add(acc[j][2 * q], (double)lo, w);
                                if constexpr (E >= 2) {
                                    const u32 hi = dw >> 16;
                                    if (!DBG(1))
                                        atomicAdd((u32*)(ldsHb + (__umul24(hi, binRowBytes) + wb)),
                                                  65536u);
                                    if (!DBG(2))
                                        acc[j][2 * q + 1] = mean_add(acc[j][2 * q + 1], (double)hi, w);
                                }
                            }
                        }
                    }
                } else {
                    // int32: same structure as uint16.  Rare path first: a lane whose vector holds a
                    // count > B (or a negative one, huge as unsigned) settles those cells here and
                    // clears them, so the common code below sees counts in [0, B] only.
                    u32 m = b.d[rb][j][0];
#pragma unroll
                    for (int q = 1; q < ND; ++q) m = max(m, b.d[rb][j][q]);
                    if (__builtin_expect(m > (u32)B, 0) && !DBG(4)) {
                        u32 ovm = 0;
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int v = (int)b.d[rb][j][e];
                            if (v < 0) {
                                atomicAdd(&sh_stat[3], 1ULL);
                                b.d[rb][j][e] = 0;
                            } else if (v > B) {
                                atomicMax(&sh_stat[4], (u64)v);
                                acc[j][e] = mean_add(acc[j][e], (double)v, w);
                                if ((long long)v < nbins) {
                                    const long long k = (long long)opaque(chunk) * E + e;
                                    atomicAdd(&a.hist[k * nbins + v], 1ULL);
                                    b.d[rb][j][e] = 0;
                                } else {
                                    ovm |= 1u << e;
                                }
                            }
                        }
                        if (__builtin_expect(ovm != 0, 0)) {
                            overflow_cells(ovm, b.d[rb][j], (long long)opaque(chunk) * E, false, 0);
#pragma unroll
                            for (int e = 0; e < E; ++e)
                                if ((ovm >> e) & 1u) b.d[rb][j][e] = 0;
                        }
                    }
                    // two counts share an LDS word; a zero count goes to the sink row
                    constexpr int EP = E >= 2 ? 2 : 1;
#pragma unroll
                    for (int e = 0; e < E; e += EP) {
                        const u32 v0 = b.d[rb][j][e];
                        const u32 v1 = EP == 2 ? b.d[rb][j][e + (EP - 1)] : 0u;
                        if (__builtin_expect((v0 | v1) != 0, 1)) {
                            const u32 wb = laneWordByte + (u32)((j * TW * PW + (e >> 1)) * CW) * 4u;
                            if (!DBG(1)) atomicAdd((u32*)(ldsHb + (__umul24(v0, binRowBytes) + wb)), 1u);
                            if (!DBG(2)) acc[j][e] = mean_add(acc[j][e], (double)v0, w);
                            if constexpr (EP == 2) {
                                if (!DBG(1))
                                    atomicAdd((u32*)(ldsHb + (__umul24(v1, binRowBytes) + wb)), 65536u);
                                if (!DBG(2))
                                    acc[j][e + (EP - 1)] = mean_add(acc[j][e + (EP - 1)], (double)v1, w);
                            }
                        }
                    }
                }
            }
        }
    };

    // Software pipeline, two register buffers: the loads of the next RB row groups are in flight
    // while the current ones are processed (the compiler places counted s_waitcnt vmcnt).
    {
        if (nsteps > 0) {
            Buf bufA, bufB;
            long long tile = tile0;
            long long k = 0;
            load_buf(bufA, tile);
            if constexpr (EARLY) {
                // the buffers refill themselves inside process_buf, two steps ahead
                load_buf(bufB, tile + tile_pitch);
                while (true) {
                    process_buf(bufA, tile, 0, tile + 2 * tile_pitch);
                    tile += tile_pitch;
                    if (++k >= nsteps) break;
                    process_buf(bufB, tile, 1, tile + 2 * tile_pitch);
                    tile += tile_pitch;
                    if (++k >= nsteps) break;
                }
            } else
            while (true) {
                load_buf(bufB, tile + tile_pitch);
                process_buf(bufA, tile, 0, 0);
                tile += tile_pitch;
                if (++k >= nsteps) break;
                load_buf(bufA, tile + tile_pitch);
                process_buf(bufB, tile, 1, 0);
                tile += tile_pitch;
                if (++k >= nsteps) break;
            }
        }
    }
    __syncthreads();

    // ---- dump the LDS histogram image as it stands (coalesced 16-byte stores); hist_reduce_kernel
    //      sums the images of all workgroups and decodes the layout.  (Flushing with global atomics
    //      instead cost ~0.5 ms at K=1000: 18M scattered 8-byte atomics, profiles/r1_notes.md.)
    {
        u32x4* dst = (u32x4*)(a.hist_part + (long long)blockIdx.x * a.part_words);
        const u32x4* src = (const u32x4*)lds;
        const int n4 = (int)(a.part_words >> 2);
        for (int i = tid; i < n4; i += BLOCK) dst[i] = src[i];
    }

    // ---- mean accumulators: lanes -> LDS -> fixed-order sum over waves (and side-by-side rows) ----
    double* ldsD = (double*)lds;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) ldsD[tid * E + e] = acc[j][e];
        __syncthreads();
        for (int t = tid; t < CC * TW * E; t += BLOCK) {
            const int cc = t / E, e = t - cc * E;
            if constexpr (TW > 1) {
                // cc = 64 * (place in the team) + lane: the sum runs over the teams of the workgroup
                const int ti = cc >> 6, ln = cc & 63;
                const int chunk = ln + 64 * (ti + TW * j);
                if (chunk < C && (long long)chunk * E + e < a.K) {
                    double sum = 0.0;
                    for (int w = 0; w < WR; ++w)
                        sum = __dadd_rn(sum, ldsD[((w * TW + ti) * 64 + ln) * E + e]);
                    a.part[(long long)blockIdx.x * a.K + (long long)chunk * E + e] = sum;
                }
            } else {
                const int chunk = MULTI ? cc + CL * j : cc + 64 * j;
                if (chunk < C && (long long)chunk * E + e < a.K) {  // (SUB > 1: a.K = KC columns)
                    double sum = 0.0;
                    for (int w = 0; w < W; ++w)
                        for (int rr = 0; rr < R; ++rr)
                            for (int sr = 0; sr < SUB; ++sr)
                                sum = __dadd_rn(sum, ldsD[(w * 64 + rr * CL + cc) * E + sr * KC + e]);
                    a.part[(long long)blockIdx.x * a.K + (long long)chunk * E + e] = sum;
                }
            }
        }
    }

    // ---- status ------------------------------------------------------------------------------
    // (the barriers of the accumulator exchange above order the LDS counters before these reads)
    if (tid == 0) atomicAdd(&a.status[0], (u64)SUB * (inter ? (blockIdx.x == 0 ? (u64)a.N : 0ULL) : (u64)nrows));
    if (tid >= 1 && tid <= 3 && sh_stat[tid]) atomicAdd(&a.status[tid], sh_stat[tid]);
    if (tid >= 4 && tid <= 5 && sh_stat[tid]) atomicMax(&a.status[tid], sh_stat[tid]);
    if (tid == 7 && sh_stat[7]) atomicAdd(&a.status[7], sh_stat[7]);  // a team wait that timed out
}

constexpr int RED_SLICES = 16;  // interleaved slices of workgroups per output (threadIdx.y)

// init_acc[k] += sum over workgroups of part[g][k], in a fixed order: RED_SLICES interleaved slices
// of g per column, combined in slice order through LDS.
__device__ void init_reduce_block(long long block, const double* part, long long G, long long K,
                                  double* init_acc) {
    __shared__ double sh[RED_SLICES][64];
    const long long k = block * 64 + threadIdx.x;
    double s = 0.0;
    if (k < K)
        for (long long g = threadIdx.y; g < G; g += RED_SLICES) s = __dadd_rn(s, part[g * K + k]);
    sh[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && k < K) {
        double t = sh[0][threadIdx.x];
#pragma unroll
        for (int y = 1; y < RED_SLICES; ++y) t = __dadd_rn(t, sh[y][threadIdx.x]);
        init_acc[k] = __dadd_rn(init_acc[k], t);
    }
}

// Sums the LDS histogram images of all workgroups and adds them into hist / histV.  One thread
// column per image word (two packed 16-bit counters), RED_SLICES interleaved slices of workgroups
// per word.
// Every (k, v) bin belongs to exactly one word, so the final += needs no atomics.
struct ReduceArgs {
    const u32* hist_part;
    long long part_words, G;
    u64* hist;
    long long nbins;
    u64* histV;
    u64* status;
    long long K;
    int C, R, B, BRW, TB, TBpad, VC, hoff, CW, PW, E, multi, RC;
    int CL;    // side-by-side rows: lanes per row
    int wide;  // 32-bit counters, one column per word (PW words per vector); sub > 1: the vector holds
    int sub;   // `sub` rows of E / sub columns
    // the mean accumulators ride in the same launch: blocks from hist_blocks on sum part[g][k]
    long long hist_blocks;
    const double* part;
    double* init_acc;
};

__global__ __launch_bounds__(64 * RED_SLICES) void hist_reduce_kernel(ReduceArgs a) {
    // a thread column owns four consecutive image words (one 16-byte load per image); the slices'
    // partial sums meet in LDS as (hi << 32 | lo) pairs of the two packed 16-bit counters
    if ((long long)blockIdx.x >= a.hist_blocks) {  // block-uniform
        init_reduce_block((long long)blockIdx.x - a.hist_blocks, a.part, a.G, a.K, a.init_acc);
        return;
    }
    __shared__ u64 sh[RED_SLICES][64][4];
    const long long idx4 = ((long long)blockIdx.x * 64 + threadIdx.x) * 4;  // first word index in the image
    const long long used = a.hoff + (long long)(a.B + 1) * a.BRW;
    u32 lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    if (idx4 < a.part_words) {  // part_words is a multiple of 4
        const u32x4* src = (const u32x4*)(a.hist_part + idx4);
        const long long stride4 = a.part_words >> 2;
        long long g = threadIdx.y;
        // sixteen loads in flight first: with one workgroup per CU (G = 256) that is a thread's whole
        // share of the images in ONE round trip (four rounds of four before: the kernel is a chain of
        // memory latencies, 14 us for 40 MB)
        for (; g + 15 * RED_SLICES < a.G; g += 16 * RED_SLICES) {
            u32x4 w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = src[(g + q * RED_SLICES) * stride4];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                lo[0] += w[q].x & 0xffffu; hi[0] += w[q].x >> 16;
                lo[1] += w[q].y & 0xffffu; hi[1] += w[q].y >> 16;
                lo[2] += w[q].z & 0xffffu; hi[2] += w[q].z >> 16;
                lo[3] += w[q].w & 0xffffu; hi[3] += w[q].w >> 16;
            }
        }
        for (; g + 3 * RED_SLICES < a.G; g += 4 * RED_SLICES) {  // four loads in flight
            u32x4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = src[(g + q * RED_SLICES) * stride4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo[0] += w[q].x & 0xffffu; hi[0] += w[q].x >> 16;
                lo[1] += w[q].y & 0xffffu; hi[1] += w[q].y >> 16;
                lo[2] += w[q].z & 0xffffu; hi[2] += w[q].z >> 16;
                lo[3] += w[q].w & 0xffffu; hi[3] += w[q].w >> 16;
            }
        }
        for (; g < a.G; g += RED_SLICES) {
            const u32x4 w = src[g * stride4];
            lo[0] += w.x & 0xffffu; hi[0] += w.x >> 16;
            lo[1] += w.y & 0xffffu; hi[1] += w.y >> 16;
            lo[2] += w.z & 0xffffu; hi[2] += w.z >> 16;
            lo[3] += w.w & 0xffffu; hi[3] += w.w >> 16;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sh[threadIdx.y][threadIdx.x][q] = (u64)lo[q] | ((u64)hi[q] << 32);
    __syncthreads();
    if (threadIdx.y != 0) return;  // one full wave goes on
    u64 maxv = 0, maxt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u64 slo = 0, shi = 0;
#pragma unroll
        for (int y = 0; y < RED_SLICES; ++y) {
            const u64 x = sh[y][threadIdx.x][q];
            slo += x & 0xffffffffu;
            shi += x >> 32;
        }
        const long long idx = idx4 + q;
        const long long vt = idx % a.TBpad;  // the bin, if idx lies in one of the VC copies of the row-total bins
        const bool isV = idx < (long long)a.VC * a.TBpad && vt < a.TB;
        const bool isH = idx >= a.hoff + a.BRW && idx < used;  // skips the weight table and the sink row
        if (isV) {  // row-total bins: plain 32-bit counters
            const u64 cnt = slo + (shi << 16);
            if (cnt) {
                if (a.VC > 1) atomicAdd(&a.histV[vt], cnt);  // VC words per bin
                else a.histV[vt] += cnt;
                maxt = maxt > (u64)vt ? maxt : (u64)vt;
            }
        } else if (isH && (slo | shi)) {
            const u32 h = (u32)(idx - a.hoff);
            const u32 v = h / (u32)a.BRW;
            const u32 wi = h - v * (u32)a.BRW;
            const u32 jp = wi / (u32)a.CW, cc = wi - jp * (u32)a.CW;
            const u32 j = jp / (u32)a.PW, p = jp - j * (u32)a.PW;
            // side-by-side rows: word cc belongs to lane cc = row slot cc / C, column chunk cc % C
            const bool lanes = a.multi && a.RC > 1;  // RC copies of a column's words
            const bool live = lanes ? cc < (u32)(a.RC * a.CL) : true;
            const u32 chunk = a.multi ? (lanes ? cc % (u32)a.CL : cc) + (u32)a.CL * j : cc + 64u * j;
            if (a.wide) {
                if (live && chunk < (u32)a.C) {
                    const long long k = a.sub > 1 ? (long long)p : (long long)chunk * a.E + p;
                    const u64 cnt = slo + (shi << 16);
                    if (lanes) atomicAdd(&a.hist[k * a.nbins + v], cnt);
                    else a.hist[k * a.nbins + v] += cnt;
                    maxv = maxv > (u64)v ? maxv : (u64)v;
                }
            } else if (live && chunk < (u32)a.C) {
                const long long k0 = (long long)chunk * a.E + (a.E >= 2 ? 2 * p : 0);
                if (lanes) {  // RC words per bin
                    if (slo) atomicAdd(&a.hist[k0 * a.nbins + v], slo);
                    if (shi) atomicAdd(&a.hist[(k0 + 1) * a.nbins + v], shi);
                } else {
                    if (slo) a.hist[k0 * a.nbins + v] += slo;
                    if (shi) a.hist[(k0 + 1) * a.nbins + v] += shi;
                }
                maxv = maxv > (u64)v ? maxv : (u64)v;
            }
        }
    }
    maxv = dme_wave_max_u64(maxv);
    maxt = dme_wave_max_u64(maxt);
    if (threadIdx.x == 0) {
        if (maxv) atomicMax(&a.status[4], maxv);
        if (maxt) atomicMax(&a.status[5], maxt);
    }
}

// ---- K = 2 (pairs of counts, the coin-flip shape) ------------------------------------------------
// uint16 rows of two columns back to back, every count below nbins, nbins^2 cells small enough for
// LDS: the pass counts the PAIRS (x0, x1) - one LDS add per row, no row total, no weight, no
// accumulator - and everything the estimation needs is read off the pair counts afterwards:
//   hist[0][v] = sum over x1 of pairs[v][x1],  hist[1][v] = sum over x0 of pairs[x0][v],
//   histV[t]   = sum over x0 + x1 == t,        init_acc[k] = sum of pairs[x0][x1] * x_k / (1 + x0 + x1)
// (the last in a fixed order: bit-reproducible).  A 16-byte vector holds four rows.  Rows with a
// count of nbins or more (a data error that status[1] reports) are settled one by one with global
// atomics.
struct PairArgs {
    const u32x4* X;
    long long NV;          // vectors of four rows
    int B1;                // cells per coordinate = nbins
    int CP;                // copies of the cells (lane % CP): equal rows in one wave add to several words
    int cstride;           // words from one copy to the next
    long long part_words;  // words of an image, a multiple of 4
    u32* hist_part;
    u64* pairs;            // [B1 * B1] summed over the images
    long long G, steps;
    u64* hist;
    long long nbins;
    u64* histV;
    long long tbins;
    double* init_acc;
    u64* status;
    u64* overflow;
};

constexpr int PAIR_BLOCK = 1024, PAIR_RB = 4;
constexpr int kPairLdsBytes = 150 * 1024;

// (int32 rows: a negative count is reported in status[3] and counts as zero, as in the general kernel)
__device__ __noinline__ void pair_odd_row(const PairArgs& a, u32 x0, u32 x1, u64* sh_stat) {
    if ((int)x0 < 0) {
        atomicAdd(&sh_stat[3], 1ULL);
        x0 = 0;
    }
    if ((int)x1 < 0) {
        atomicAdd(&sh_stat[3], 1ULL);
        x1 = 0;
    }
    if ((x0 | x1) == 0) return;
    const u64 total = (u64)x0 + (u64)x1;
    const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 v = k ? x1 : x0;
        if (v == 0) continue;
        atomicMax(&sh_stat[4], (u64)v);
        if ((long long)v < a.nbins) atomicAdd(&a.hist[(long long)k * a.nbins + v], 1ULL);
        else if (!dme_overflow_put(a.overflow, (u64)k, (u64)v, 1ULL)) atomicAdd(&sh_stat[1], 1ULL);
        atomicAdd(&a.init_acc[k], __dmul_rn((double)v, w));
    }
    atomicMax(&sh_stat[5], (u64)total);
    if ((long long)total < a.tbins) atomicAdd(&a.histV[total], 1ULL);
    else atomicAdd(&sh_stat[2], 1ULL);
}

// I32: int32 rows, two to a vector
template <bool I32>
__global__ __launch_bounds__(PAIR_BLOCK) void stats_pair_kernel(PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    __shared__ u64 sh_stat[8];
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)a.part_words; i += PAIR_BLOCK) lds[i] = 0;
    if (tid < 8) sh_stat[tid] = 0;
    __syncthreads();
    const u32 B1 = (u32)a.B1;
    u32* cells = lds + (u32)(tid & (a.CP - 1)) * (u32)a.cstride;
    auto load = [&](u32x4 (&v)[PAIR_RB], long long step) {
        const long long base = (step * gridDim.x + blockIdx.x) * (long long)(PAIR_BLOCK * PAIR_RB);
#pragma unroll
        for (int rb = 0; rb < PAIR_RB; ++rb) {
            const long long g = base + rb * PAIR_BLOCK + tid;
            v[rb] = g < a.NV ? __builtin_nontemporal_load(a.X + g) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto one = [&](u32 dw) {
        if (dw == 0) return;  // an empty row
        const u32 x0 = dw & 0xffffu, x1 = dw >> 16;
        if (__builtin_expect(max(x0, x1) < B1, 1)) atomicAdd(&cells[__umul24(x0, B1) + x1], 1u);
        else pair_odd_row(a, x0, x1, sh_stat);
    };
    auto one32 = [&](u32 x0, u32 x1) {
        if ((x0 | x1) == 0) return;  // an empty row
        if (__builtin_expect(max(x0, x1) < B1, 1)) atomicAdd(&cells[__umul24(x0, B1) + x1], 1u);
        else pair_odd_row(a, x0, x1, sh_stat);  // (a negative count reads as a huge one)
    };
    auto process = [&](u32x4 (&v)[PAIR_RB]) {
#pragma unroll
        for (int rb = 0; rb < PAIR_RB; ++rb) {
            if constexpr (I32) {
                one32(v[rb].x, v[rb].y);
                one32(v[rb].z, v[rb].w);
            } else {
                one(v[rb].x);
                one(v[rb].y);
                one(v[rb].z);
                one(v[rb].w);
            }
        }
    };
    if (a.steps > 0) {
        u32x4 va[PAIR_RB], vb[PAIR_RB];
        long long k = 0;
        load(va, 0);
        while (true) {
            load(vb, k + 1);
            process(va);
            if (++k >= a.steps) break;
            load(va, k + 1);
            process(vb);
            if (++k >= a.steps) break;
        }
    }
    __syncthreads();
    {
        u32x4* dst = (u32x4*)(a.hist_part + (long long)blockIdx.x * a.part_words);
        const u32x4* src = (const u32x4*)lds;
        for (int i = tid; i < (int)(a.part_words >> 2); i += PAIR_BLOCK) dst[i] = src[i];
    }
    if (tid == 0 && blockIdx.x == 0) atomicAdd(&a.status[0], (I32 ? 2ULL : 4ULL) * (u64)a.NV);
    if (tid >= 1 && tid <= 3 && sh_stat[tid]) atomicAdd(&a.status[tid], sh_stat[tid]);
    if (tid >= 4 && tid <= 5 && sh_stat[tid]) atomicMax(&a.status[tid], sh_stat[tid]);
}

// pairs[c] = sum over the images and their copies; block (64, RED_SLICES): a thread column per cell.
// The G * CP words of a cell are taken eight at a time (eight loads in flight: with one at a time a
// small input - BASELINE configs[1], 122 images - spent 20 us here on memory latency alone).
__global__ __launch_bounds__(64 * RED_SLICES) void pair_sum_kernel(PairArgs a) {
    __shared__ u64 sh[RED_SLICES][64];
    const long long ncell = (long long)a.B1 * a.B1;
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    u64 s = 0;
    if (c < ncell) {
        const long long total = a.G * a.CP;  // word j of the cell: image j / CP, copy j % CP (CP is a power of two)
        const int cpm = a.CP - 1, cps = __ffs(a.CP) - 1;
        auto word = [&](long long j) -> u32 {
            return a.hist_part[(j >> cps) * a.part_words + (long long)(j & cpm) * a.cstride + c];
        };
        long long j = threadIdx.y;
        for (; j + 7 * RED_SLICES < total; j += 8 * RED_SLICES) {
            u32 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = word(j + u * RED_SLICES);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += x[u];
        }
        for (; j < total; j += RED_SLICES) s += word(j);
    }
    sh[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < ncell) {
        u64 t = 0;
#pragma unroll
        for (int y = 0; y < RED_SLICES; ++y) t += sh[y][threadIdx.x];
        a.pairs[c] = t;
    }
}

// one workgroup: the marginals of the pair counts.  ONE pass over the cells: a cell (x0, x1) adds its rows
// to three LDS tables - column 0's count x0, column 1's count x1, the row total x0 + x1 - and its share to
// the mean accumulators (per thread in cell order, then a tree: as before, the same bits); the tables
// then go to hist / histV.  (Three passes before, a thread walking a whole row or column of the pair
// table by itself: 23 us for 61 x 61 cells.)
__global__ __launch_bounds__(1024) void pair_marginals_kernel(PairArgs a) {
    __shared__ double sh0[1024], sh1[1024];
    __shared__ u64 h0[1024], h1[1024], ht[2048];  // B1 <= 1024 (stats_pair_run)
    __shared__ u64 sh_max[2];
    const int tid = threadIdx.x, B1 = a.B1;
    if (tid < 2) sh_max[tid] = 0;
    h0[tid] = 0;
    h1[tid] = 0;
    ht[tid] = 0;
    ht[tid + 1024] = 0;
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    for (int c = tid; c < B1 * B1; c += 1024) {
        const u64 n = a.pairs[c];
        if (n) {
            const int x0 = c / B1, x1 = c - x0 * B1;
            atomicAdd(&h0[x0], n);
            atomicAdd(&h1[x1], n);
            atomicAdd(&ht[x0 + x1], n);
            const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)(x0 + x1)));
            s0 = __dadd_rn(s0, __dmul_rn(__dmul_rn((double)n, (double)x0), w));
            s1 = __dadd_rn(s1, __dmul_rn(__dmul_rn((double)n, (double)x1), w));
        }
    }
    sh0[tid] = s0;
    sh1[tid] = s1;
    __syncthreads();
    for (int v = 1 + tid; v < B1; v += 1024) {
        const u64 c0 = h0[v], c1 = h1[v];
        if (c0) a.hist[v] += c0;
        if (c1) a.hist[a.nbins + v] += c1;
        if (c0 | c1) atomicMax(&sh_max[0], (u64)v);
    }
    for (int t = 1 + tid; t <= 2 * (B1 - 1); t += 1024) {
        const u64 cnt = ht[t];
        if (cnt) {
            atomicMax(&sh_max[1], (u64)t);
            if ((long long)t < a.tbins) a.histV[t] += cnt;
            else atomicAdd(&a.status[2], cnt);
        }
    }
    for (int d = 512; d > 0; d >>= 1) {
        if (tid < d) {
            sh0[tid] = __dadd_rn(sh0[tid], sh0[tid + d]);
            sh1[tid] = __dadd_rn(sh1[tid], sh1[tid + d]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.init_acc[0] = __dadd_rn(a.init_acc[0], sh0[0]);
        a.init_acc[1] = __dadd_rn(a.init_acc[1], sh1[0]);
        if (sh_max[0]) atomicMax(&a.status[4], sh_max[0]);
        if (sh_max[1]) atomicMax(&a.status[5], sh_max[1]);
    }
}

// Returns DME_OK, an error, or -1 when the shape is not this pass's (too many bins, workspace).
int stats_pair_run(const void* X, bool i32, long long NV, u64* hist, long long nbins, u64* histV,
                   long long tbins, double* init_acc, u64* status, u64* overflow, void* workspace,
                   long long workspace_bytes, hipStream_t s) {
    if (nbins < 2 || nbins > 1024) return -1;
    const long long ncell = nbins * nbins;
    long long cstride = (ncell + 3) & ~3LL;
    cstride += 1;  // odd: the copies of a cell fall into different banks
    int CP = 64;
    while (CP > 1 && CP * cstride * 4 > kPairLdsBytes) CP >>= 1;
    if (CP * cstride * 4 > kPairLdsBytes) return -1;
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    const long long per = (long long)PAIR_BLOCK * PAIR_RB;
    long long G = (NV + per - 1) / per;
    if (G > cus) G = cus;
    if (G < 1) G = 1;
    if (NV / G >= (1LL << 29)) return -1;  // 32-bit cells
    if (i32 && nbins > 16384) return -1;   // (the cell index is a 24-bit product)
    PairArgs a{};
    a.X = (const u32x4*)X;
    a.NV = NV;
    a.B1 = (int)nbins;
    a.CP = CP;
    a.cstride = (int)cstride;
    a.part_words = (CP * cstride + 3) & ~3LL;
    const long long img_bytes = (G * a.part_words * 4 + 255) & ~255LL;
    if (!workspace || workspace_bytes < img_bytes + ncell * 8) return -1;
    a.hist_part = (u32*)workspace;
    a.pairs = (u64*)((char*)workspace + img_bytes);
    a.G = G;
    a.steps = (NV + G * per - 1) / (G * per);
    a.hist = hist; a.nbins = nbins; a.histV = histV; a.tbins = tbins; a.init_acc = init_acc; a.status = status; a.overflow = overflow;
    static std::atomic<bool> attr_set[64];  // zero-initialised; racing first calls both set the attribute
    int dev = 0;
    DME_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)stats_pair_kernel<false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kPairLdsBytes + 1024));
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)stats_pair_kernel<true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kPairLdsBytes + 1024));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    if (i32)
        hipLaunchKernelGGL(stats_pair_kernel<true>, dim3((unsigned)G), dim3(PAIR_BLOCK), (size_t)(a.part_words * 4), s, a);
    else
        hipLaunchKernelGGL(stats_pair_kernel<false>, dim3((unsigned)G), dim3(PAIR_BLOCK), (size_t)(a.part_words * 4), s, a);
    hipLaunchKernelGGL(pair_sum_kernel, dim3((unsigned)((ncell + 63) / 64)), dim3(64, RED_SLICES), 0, s, a);
    hipLaunchKernelGGL(pair_marginals_kernel, dim3(1), dim3(1024), 0, s, a);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

constexpr int kLdsBudgetBytes = 156 * 1024;  // of the 160 KiB per CU; one workgroup per CU
constexpr int kMaxJ = 8;      // vectors per lane and row
constexpr int kMaxJPack = 2;  // the same for the packed int32 kernel inside a team
constexpr long long kRowsPerWgMax = 60000;
constexpr long long kSubMinRows = 1 << 16;  // rows inside a vector (K = 2, 4): from this many rows on

struct Plan {
    int E, J, TW, C, R, B, BRW, TB, TBpad, WT, hoff, block, lds_bytes, rows_per_wg, steps, adjacent, VC, CW, Vw, RC, CL, tail;
    long long grid, part_words;
    bool multi;
    bool pack;  // int32 rows through the packed kernel (E = 8 columns per chunk)
    bool wide;  // 32-bit LDS counters (side-by-side uint16 rows, very many of them)
    bool early; // packed int32: buffers refilled as soon as their counts are packed (see the kernel)
    int sub;    // rows inside a 16-byte vector (K = 2 or 4, rows back to back), else 1
};

// `padded`: the caller's rows are 16-byte aligned, their pitch covers K rounded up to 8 columns and
// the columns from K on hold zeros: the vectors may then run past K (no caller at present).
// `sub` > 1: uint16 rows of K = 8 / sub columns that lie back to back from a 16-byte-aligned address:
// N counts vectors of `sub` rows.
int plan_dense(int elem_bytes, const void* X, long long N, long long K, long long ld,
               long long nbins, long long tbins, bool padded, Plan* pl, int sub = 1) {
    int E = 16 / elem_bytes;
    const uintptr_t addr = (uintptr_t)X;
    if (addr % elem_bytes) return dme_set_error(DME_ERR_INVALID_ARG, "X is not element-aligned");
    if (padded) {
        if (addr % 16 || (ld * elem_bytes) % 16 || ld < ((K + 7) & ~7LL))
            return dme_set_error(DME_ERR_INTERNAL, "padded rows: bad pitch");
    } else if (sub > 1) {
        if (elem_bytes != 2 || K * sub != 8 || ld != K || addr % 16)
            return dme_set_error(DME_ERR_INTERNAL, "rows inside a vector: bad layout");
    } else if (K < E) {
        // rows shorter than a vector: narrower loads
        while (E > 1 && (K % E || ld % E || (addr % ((uintptr_t)E * elem_bytes)))) E >>= 1;
    }
    // (otherwise full vectors whatever the pitch and the alignment: see load_chunk and `tail`)
    // int32 rows in chunks of 8 columns (rows of fewer than 64 chunks side by side, like uint16 rows)
    const bool pack = elem_bytes == 4 && E == 4 && K >= 8;
    if (pack) E = 8;
    const long long C = sub > 1 ? 1 : (K + E - 1) / E;
    Plan p{};
    p.tail = (!padded && sub == 1 && K >= E) ? (int)(K % E) : 0;
    p.sub = sub;
    p.pack = pack;
    p.E = E;
    p.C = (int)C;
    p.TW = 1;
    p.multi = C < 64;
    // vectors per lane that one wave takes; longer rows go to teams of 2, 4 or 8 waves (full
    // vectors of 8 columns only)
    const int jmax = pack ? kMaxJPack : kMaxJ;
    const long long jtot = (C + 63) / 64;
    p.CL = 64;
    if (p.multi) {
        p.J = 1;
        p.R = (int)(64 / C);
        p.CL = (int)C;
    }
    // Lanes in use: a row of C vectors on its own lanes fills C * R of 64 (K = 200: 25 * 2), and one
    // of 65 vectors fills 65 of 128 lane slots.  uint16 rows in full vectors may instead take
    // CL = ceil(C / J) lanes with J vectors each, R = 64 / CL rows side by side (K = 200: J = 3,
    // 9 lanes, 7 rows: 175 of 192 slots).
    // The share of a wave's work that is of use, for rows on CL lanes with J vectors each: a vector is
    // four steps of a dword (two counts) per lane, and a step is skipped when no lane of the wave has
    // counts there - the dwords beyond a row's end, when all lanes hold the row's last vector at the
    // same j (K = 10: one lane per row, the second vector one dword: 5 steps per row instead of the 8
    // of two lanes per row).
    auto useful = [&](long long j, long long cl) {
        const long long taild = p.tail ? (p.tail + 1) / 2 : 4;  // dwords of a row's last vector
        long long steps = 0;
        for (long long jj = 0; jj < j; ++jj) {
            long long m = 0;
            for (long long cc = 0; cc < cl; ++cc) {
                const long long v = cc + cl * jj;
                const long long d = v < C - 1 ? 4 : (v == C - 1 ? taild : 0);
                if (d > m) m = d;
            }
            steps += m;
        }
        const long long rr = 64 / cl;
        const double dwords = 4.0 * (double)(C - 1) + (double)taild;  // of one row
        return steps ? dwords * (double)rr / (64.0 * (double)steps) : 0.0;
    };
    if (E == 8 && sub == 1 && C <= (pack ? 2 : 5) * 64) {
        // (the packed int32 kernel counts whole vectors: its two-vector variant is the slower one)
        double best = !p.multi ? (double)C / (64.0 * (double)jtot)
                      : pack   ? (double)(C * p.R) / 64.0
                               : useful(1, C);
        for (int j = 1; j <= (pack ? 2 : 5); ++j) {
            const long long cl = (C + j - 1) / j;
            if (cl > 64 || (p.multi && j == 1)) continue;
            // a lane per row with three or more vectors: every load instruction of the wave touches
            // every cache line of its rows (K = 20: 2.73 TB/s against 3.92 on three lanes per row)
            if (cl == 1 && j > 2) continue;
            const long long rr = 64 / cl;
            // 4 and 5 vectors per lane run fewer threads
            const double u = pack ? (double)(C * rr) / (64.0 * j)
                                  : useful(j, cl) * (j == 4 ? 0.95 : j == 5 ? 0.85 : 1.0);
            // (the packed int32 kernel of two vectors per lane runs 768 threads: K = 100 measured 4.45
            // TB/s with one vector per lane on 52 lanes against 3.67 with two on 63)
            // rows of 33 to 63 vectors fill one wave with J = 1: two of them side by side with two vectors
            // per lane use the same share of the lanes and halve the per-row work of a wave (K = 400: 4.16
            // -> 4.85 TB/s, K = 448: 4.63 -> 5.53, K = 480: 4.98 -> 5.68)
            const bool pair_up = !pack && j == 2 && p.multi && p.J == 1 && p.R == 1 && u >= best;
            if (u > best * (pack ? 1.3 : 1.08) || pair_up) {
                best = u;
                p.multi = true;
                p.J = j;
                p.CL = (int)cl;
                p.R = (int)rr;
            }
        }
    }
    if (p.multi) {
    } else if (elem_bytes == 2 && E == 8 && (jtot == 6 || jtot == 8 || (jtot >= 13 && jtot <= 16))) {
        // two (four) waves of 3 or 4 vectors per lane at 1024 / 768 threads rather than one wave of 6
        // or 8 (two of 8) at 512: K = 3000: 4.91 -> 5.69 TB/s, K = 4096: 5.50 -> 5.65, K = 8192: 5.20 ->
        // 5.38; teams of 8 waves measured slower than what they would replace
        p.TW = jtot > 8 ? 4 : 2;
        p.J = jtot == 6 ? 3 : 4;
        p.R = 1;
    } else if (jtot > jmax && E == 8 && jtot <= (pack ? 16LL : 8LL) * jmax) {
        int TW = 2;
        while (TW * (long long)jmax < jtot) TW *= 2;
        p.TW = TW;
        p.J = (int)((jtot + TW - 1) / TW);
        if (p.J <= jmax / 2) p.J = jmax / 2 + 1;  // the team kernels exist for the upper half of J
        p.R = 1;
    } else {
        // any number of vectors per lane up to kMaxJ (K = 3000 uint16: 6, not 8 with a quarter of the
        // lanes' loads empty)
        if (jtot > kMaxJ)
            return dme_set_error(DME_ERR_UNSUPPORTED,
                                 "dense stats: K=%lld in %d-element vectors needs %lld vectors per "
                                 "lane (at most %d, or %d of 8 elements): use the CSR entry",
                                 K, E, jtot, kMaxJ, (pack ? 16 : 8) * jmax);
        p.J = (int)jtot;
        p.R = 1;
    }
    // Packed int32 rows: two buffers of loads in flight wherever one leaves the memory pipe short - teams of
    // four waves or more (the widest rows: 3.6 -> 5.8 TB/s at K = 16 384), rows that end inside a vector
    // (K = 10: 3.4 -> 4.4, K = 1001: 4.1 -> 4.9) and shapes that leave a tenth or more of the lane slots
    // empty (K = 600: 3.8 -> 4.7, K = 1536: 4.4 -> 5.1).  Shapes that fill their waves measured 4-7 %
    // slower with it (K = 8, 16, 512, 1000, 1024) and run without.
    if (pack) {
        const double slots = p.multi ? 64.0 * p.J : 64.0 * p.J * p.TW;
        const double fill = (double)(C * (p.multi ? p.R : 1)) / slots;
        p.early = p.TW >= 4 || p.tail != 0 || fill < 0.9;
    }
    int PW = E >= 2 ? E / 2 : 1;
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    // many side-by-side rows: a word per lane (see the kernel) - when there are rows enough to repay
    // the larger images that every workgroup dumps and hist_reduce_kernel sums (~20 us)
    const bool many_rows = N >= 4000000;
    int RC = (p.multi && p.R >= 16 && many_rows) ? p.R : 1;
    p.block = stats_block(p.J, E, elem_bytes, p.TW, p.multi);
    long long TB = tbins < 2048 ? tbins : 2048;
    const long long TBpad = (TB + 3) & ~3LL;
    const long long WT = ((TB < 1024 ? TB : 1024) + 1) & ~1LL;  // even: keeps hoff a multiple of 4
    // copies of the row-total bins for side-by-side rows: one per row slot, within 4096 words
    long long VC = 1;
    if (p.multi && many_rows)
        while (VC < 64 && VC * 2 <= p.R && VC * 2 * (TBpad + 1) <= 4096 + 64) VC *= 2;
    const long long Vstride = VC > 1 ? TBpad + 1 : TBpad;  // odd: spreads equal totals over the banks
    const long long Vw = (VC * Vstride + 3) & ~3LL;
    const long long hoff = Vw + 2 * WT;
    long long avail = kLdsBudgetBytes / 4 - hoff - 4;
    // ... as many copies as leave room for all the count bins (halved until they fit)
    auto words = [&](int rc) { return !p.multi ? 64 : (rc == 1 ? p.CL : (rc == p.R ? 64 : rc * p.CL)); };
    // 32-bit counters (see the kernel) when the rows would otherwise take several rounds of
    // workgroups and every bin still fits in LDS; always for rows inside a vector
    if (p.multi && (elem_bytes == 2 || pack) && E == 8 && !padded &&
        (sub > 1 || N > (long long)cus * kRowsPerWgMax)) {
        const int PWw = E / sub;
        int rcw = RC;
        while (rcw > 1 && avail / ((long long)p.J * PWw * words(rcw)) - 1 < nbins - 1) rcw /= 2;
        if (sub > 1 || avail / ((long long)p.J * PWw * words(rcw)) - 1 >= nbins - 1) {
            p.wide = true;
            PW = PWw;
            RC = rcw;
        }
    }
    while (RC > 1 && avail / ((long long)p.J * p.TW * PW * words(RC)) - 1 < nbins - 1) RC /= 2;
    p.RC = RC;
    p.CW = words(RC);
    p.BRW = p.J * p.TW * PW * p.CW;
    long long B = avail / p.BRW - 1;
    if (B > nbins - 1) B = nbins - 1;
    if (B > 65535) B = 65535;
    if (B < 0) B = 0;
    p.B = (int)B;
    p.TB = (int)TB;
    p.TBpad = (int)Vstride;
    p.VC = (int)VC;
    p.Vw = (int)Vw;
    p.WT = (int)WT;
    p.hoff = (int)hoff;
    p.part_words = (hoff + (B + 1) * (long long)p.BRW + 3) & ~3LL;
    long long lds = p.part_words * 4;
    const long long acc_bytes = (long long)p.block * E * 8;  // mean-accumulator exchange
    if (lds < acc_bytes) lds = acc_bytes;
    p.lds_bytes = (int)((lds + 15) & ~15LL);
    // one workgroup per CU per round; a workgroup handles at most 65535 rows (16-bit LDS counters):
    // 60000 leaves room for the rounding of the interleaved mapping (whole loop steps)
    long long rounds = (N + (long long)cus * kRowsPerWgMax - 1) / ((long long)cus * kRowsPerWgMax);
    if (rounds < 1 || p.wide) rounds = 1;
    if (p.wide && N / cus * sub >= (1LL << 31))
        return dme_set_error(DME_ERR_UNSUPPORTED, "dense stats: %lld rows are more than the counters hold", N * sub);
    long long grid = (long long)cus * rounds;
    long long rpw = (N + grid - 1) / grid;
    const long long min_rows = 64LL * p.R;
    if (rpw < min_rows) rpw = min_rows;
    grid = (N + rpw - 1) / rpw;
    if (grid < 1) grid = 1;
    p.rows_per_wg = (int)rpw;
    p.grid = grid;
    // interleaved mapping: S row groups per workgroup and loop step (S as in the kernel: waves x
    // register-buffer depth), steps = ceil(groups / (grid * S)); a workgroup still counts at most
    // 65535 rows
    {
        const int W = p.block / 64 / p.TW;
        const int RB = stats_rb(p.J, E, elem_bytes, p.TW, p.multi);
        const long long S = (long long)W * RB;
        const long long groups = (N + p.R - 1) / p.R;
        long long steps = (groups + grid * S - 1) / (grid * S);
        if (!p.wide && steps * S * p.R > 65535) steps = 0;  // contiguous slabs (rows_per_wg <= 60000 each)
        p.steps = (int)steps;
        if (pack) p.steps = 0;
        // rows that share cache lines with their neighbours (pitch not a multiple of 128 bytes)
        p.adjacent = (!p.multi && (ld * elem_bytes) % 128 != 0) ? 1 : 0;
        if (p.steps == 0) p.grid = (N + rpw - 1) / rpw;
    }
    *pl = p;
    return DME_OK;
}

template <typename T, int E, int J, bool MULTI, bool PACK, int TW, bool SCAN, bool WIDE, int SUB, bool TAIL>
int launch_dense_t(const StatsArgs& a, const Plan& p, hipStream_t s);

template <typename T, int E, int J, bool MULTI, bool PACK = false, int TW = 1, bool SCAN = false,
          bool WIDE = false, int SUB = 1>
int launch_dense(const StatsArgs& a, const Plan& p, hipStream_t s) {
    if constexpr (E * sizeof(T) >= 16 && SUB == 1) {
        if (p.tail) return launch_dense_t<T, E, J, MULTI, PACK, TW, SCAN, WIDE, SUB, true>(a, p, s);
    }
    return launch_dense_t<T, E, J, MULTI, PACK, TW, SCAN, WIDE, SUB, false>(a, p, s);
}

template <typename T, int E, int J, bool MULTI, bool PACK, int TW, bool SCAN, bool WIDE, int SUB, bool TAIL,
          bool EARLY>
int launch_dense_e(const StatsArgs& a, const Plan& p, hipStream_t s);

template <typename T, int E, int J, bool MULTI, bool PACK, int TW, bool SCAN, bool WIDE, int SUB, bool TAIL>
int launch_dense_t(const StatsArgs& a, const Plan& p, hipStream_t s) {
    if constexpr (PACK) {
        if (p.early) return launch_dense_e<T, E, J, MULTI, PACK, TW, SCAN, WIDE, SUB, TAIL, true>(a, p, s);
    }
#ifdef DME_U16_EARLY_ALL  // experiment: every uint16 kernel with three buffers
    if constexpr (sizeof(T) == 2) return launch_dense_e<T, E, J, MULTI, PACK, TW, SCAN, WIDE, SUB, TAIL, true>(a, p, s);
#endif
    return launch_dense_e<T, E, J, MULTI, PACK, TW, SCAN, WIDE, SUB, TAIL, false>(a, p, s);
}

template <typename T, int E, int J, bool MULTI, bool PACK, int TW, bool SCAN, bool WIDE, int SUB, bool TAIL,
          bool EARLY>
int launch_dense_e(const StatsArgs& a, const Plan& p, hipStream_t s) {
    auto kern = stats_dense_kernel<T, E, J, MULTI, PACK, TW, SCAN, WIDE, SUB, TAIL, EARLY>;
    // allow the full 160 KiB of dynamic LDS (the default cap is 64 KiB); once per kernel and device
    static std::atomic<bool> attr_set[64];  // zero-initialised; racing first calls both set the attribute
    int dev = 0;
    DME_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kLdsBudgetBytes + 1024));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(p.block), p.lds_bytes, s, a);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

// side-by-side uint16 rows in full vectors: J vectors per lane, row totals by shuffles or by a scan,
// 16- or 32-bit counters
template <typename T, int E, int J>
int dispatch_multi(const StatsArgs& a, const Plan& p, hipStream_t s) {
    if (p.wide)
        return p.CL > 8 ? launch_dense<T, E, J, true, false, 1, true, true>(a, p, s)
                        : launch_dense<T, E, J, true, false, 1, false, true>(a, p, s);
    return p.CL > 8 ? launch_dense<T, E, J, true, false, 1, true>(a, p, s)
                    : launch_dense<T, E, J, true, false, 1, false>(a, p, s);
}

template <typename T, int E>
int dispatch_j(const StatsArgs& a, const Plan& p, hipStream_t s) {
    if constexpr (sizeof(T) == 2 && E == 8) {
        if (p.sub == 4) return launch_dense<T, E, 1, true, false, 1, false, true, 4>(a, p, s);
        if (p.sub == 2) return launch_dense<T, E, 1, true, false, 1, false, true, 2>(a, p, s);
        if (p.multi) switch (p.J) {
            case 1: return dispatch_multi<T, E, 1>(a, p, s);
            case 2: return dispatch_multi<T, E, 2>(a, p, s);
            case 3: return dispatch_multi<T, E, 3>(a, p, s);
            case 4: return dispatch_multi<T, E, 4>(a, p, s);
            case 5: return dispatch_multi<T, E, 5>(a, p, s);
        }
    }
    if (p.multi)
        return p.CL > 8 ? launch_dense<T, E, 1, true, false, 1, true>(a, p, s)
                        : launch_dense<T, E, 1, true>(a, p, s);
    switch (p.J) {
        case 1: return launch_dense<T, E, 1, false>(a, p, s);
        case 2: return launch_dense<T, E, 2, false>(a, p, s);
        case 3: return launch_dense<T, E, 3, false>(a, p, s);
        case 4: return launch_dense<T, E, 4, false>(a, p, s);
        case 5: return launch_dense<T, E, 5, false>(a, p, s);
        case 6: return launch_dense<T, E, 6, false>(a, p, s);
        case 7: return launch_dense<T, E, 7, false>(a, p, s);
        case 8: return launch_dense<T, E, 8, false>(a, p, s);
    }
    return dme_set_error(DME_ERR_INTERNAL, "bad J=%d", p.J);
}

// rows split over teams of TW waves: full vectors of 8 columns, the upper half of J
template <typename T, bool PACK, int TW>
int dispatch_team(const StatsArgs& a, const Plan& p, hipStream_t s) {
    if constexpr (PACK) {
        switch (p.J) {
            case 2: return launch_dense<T, 8, 2, false, true, TW>(a, p, s);
        }
    } else if constexpr (sizeof(T) == 2) {
        switch (p.J) {
            case 3: return launch_dense<T, 8, 3, false, false, TW>(a, p, s);
            case 4: return launch_dense<T, 8, 4, false, false, TW>(a, p, s);
            case 5: return launch_dense<T, 8, 5, false, false, TW>(a, p, s);
            case 6: return launch_dense<T, 8, 6, false, false, TW>(a, p, s);
            case 7: return launch_dense<T, 8, 7, false, false, TW>(a, p, s);
            case 8: return launch_dense<T, 8, 8, false, false, TW>(a, p, s);
        }
    }
    return dme_set_error(DME_ERR_INTERNAL, "bad team J=%d", p.J);
}

template <typename T>
int stats_dense_run(const T* X, long long N, long long K, long long ld, u64* hist, long long nbins,
                    u64* histV, long long tbins, double* init_acc, u64* status, u64* overflow, void* workspace,
                    long long workspace_bytes, void* stream, bool padded) {
    // uint16 rows of 2 or 4 columns back to back: whole rows inside 16-byte vectors, the few rows
    // that do not fill a vector afterwards on their own
    int sub = 1;
    if (sizeof(T) == 2 && !padded && (K == 2 || K == 4) && ld == K && ((uintptr_t)X % 16) == 0 &&
        N >= kSubMinRows)
        sub = (int)(8 / K);
    // K = 2 (rows back to back, both element types): pair counts when the bins allow
    if (!padded && K == 2 && ld == 2 && ((uintptr_t)X % 16) == 0 && N >= kSubMinRows) {
        const long long rpv = 16 / (2 * (long long)sizeof(T));  // rows per vector
        int rc = stats_pair_run(X, sizeof(T) == 4, N / rpv, hist, nbins, histV, tbins, init_acc,
                                status, overflow, workspace, workspace_bytes, (hipStream_t)stream);
        if (rc > 0) return rc;
        if (rc == DME_OK) {
            if (N % rpv)
                return stats_dense_run<T>(X + (N / rpv) * rpv * 2, N % rpv, K, ld, hist, nbins, histV, tbins,
                                          init_acc, status, overflow, workspace, workspace_bytes, stream, false);
            return DME_OK;
        }
    }
    const long long Nall = N;
    if (sub > 1) N = Nall / sub;
    Plan p;
    int rc = plan_dense((int)sizeof(T), X, N, K, ld, nbins, tbins, padded, &p, sub);
    if (rc) return rc;
    const long long init_bytes = (p.grid * K * 8 + 255) & ~255LL;
    const long long need = init_bytes + p.grid * p.part_words * 4;
    if (!workspace || workspace_bytes < need)
        return dme_set_error(DME_ERR_WORKSPACE, "stats workspace: need %lld bytes, got %lld", need,
                             workspace_bytes);
    DME_REQUIRE(((uintptr_t)workspace % 16) == 0, "workspace must be 16-byte aligned");
    u32* hist_part = (u32*)((char*)workspace + init_bytes);
    StatsArgs a{X, N, K, sub > 1 ? 8LL : ld, hist, nbins, histV, tbins, (double*)workspace, hist_part,
                p.part_words, status, p.rows_per_wg, p.C, p.R, p.B, p.BRW, p.TB, p.TBpad, p.VC, p.Vw, p.WT, p.hoff, p.CW, p.steps, p.adjacent, p.RC, p.CL, p.tail, overflow};
    hipStream_t s = (hipStream_t)stream;
    constexpr int EMAX = 16 / (int)sizeof(T);
    if (p.TW > 1) {
        if constexpr (sizeof(T) == 4) {
            if (!p.pack) return dme_set_error(DME_ERR_INTERNAL, "team without packing");
            switch (p.TW) {
                case 2: rc = dispatch_team<T, true, 2>(a, p, s); break;
                case 4: rc = dispatch_team<T, true, 4>(a, p, s); break;
                case 8: rc = dispatch_team<T, true, 8>(a, p, s); break;
                case 16: rc = dispatch_team<T, true, 16>(a, p, s); break;
                default: rc = dme_set_error(DME_ERR_INTERNAL, "bad TW=%d", p.TW);
            }
        } else {
            switch (p.TW) {
                case 2: rc = dispatch_team<T, false, 2>(a, p, s); break;
                case 4: rc = dispatch_team<T, false, 4>(a, p, s); break;
                case 8: rc = dispatch_team<T, false, 8>(a, p, s); break;
                default: rc = dme_set_error(DME_ERR_INTERNAL, "bad TW=%d", p.TW);
            }
        }
    } else if constexpr (sizeof(T) == 4) {
        if (p.pack && p.multi) {
            const bool scan = p.CL > 8;
            switch (p.J * 2 + (p.wide ? 1 : 0)) {
                case 2:
                    rc = scan ? launch_dense<T, 8, 1, true, true, 1, true>(a, p, s)
                              : launch_dense<T, 8, 1, true, true, 1, false>(a, p, s);
                    break;
                case 3:
                    rc = scan ? launch_dense<T, 8, 1, true, true, 1, true, true>(a, p, s)
                              : launch_dense<T, 8, 1, true, true, 1, false, true>(a, p, s);
                    break;
                case 4:
                    rc = scan ? launch_dense<T, 8, 2, true, true, 1, true>(a, p, s)
                              : launch_dense<T, 8, 2, true, true, 1, false>(a, p, s);
                    break;
                case 5:
                    rc = scan ? launch_dense<T, 8, 2, true, true, 1, true, true>(a, p, s)
                              : launch_dense<T, 8, 2, true, true, 1, false, true>(a, p, s);
                    break;
                default: rc = dme_set_error(DME_ERR_INTERNAL, "bad J=%d", p.J);
            }
        } else if (p.pack) {
            switch (p.J) {
                case 1: rc = launch_dense<T, 8, 1, false, true>(a, p, s); break;
                case 2: rc = launch_dense<T, 8, 2, false, true>(a, p, s); break;
                default: rc = dme_set_error(DME_ERR_INTERNAL, "bad J=%d", p.J);
            }
        }
    }
    if (!p.pack && p.TW == 1) switch (p.E) {
        case 1: rc = dispatch_j<T, 1>(a, p, s); break;
        case 2: rc = dispatch_j<T, 2>(a, p, s); break;
        case 4: rc = dispatch_j<T, 4>(a, p, s); break;
        case 8:
            if constexpr (EMAX >= 8) rc = dispatch_j<T, 8>(a, p, s);
            else rc = dme_set_error(DME_ERR_INTERNAL, "bad E");
            break;
        default: rc = dme_set_error(DME_ERR_INTERNAL, "bad E=%d", p.E);
    }
    if (rc) return rc;
    const int PW = p.wide ? p.E / p.sub : (p.E >= 2 ? p.E / 2 : 1);
    // (a team's image reads like a one-wave row's of TW * J vectors per lane)
    ReduceArgs ra{hist_part, p.part_words, p.grid, hist, nbins, histV, status, K,
                  p.C, p.R, p.B, p.BRW, p.TB, p.TBpad, p.VC, p.hoff, p.CW, PW, p.E,
                  p.multi ? 1 : 0, p.RC, p.CL, p.wide ? 1 : 0, p.sub};
    const long long used = p.hoff + (long long)(p.B + 1) * p.BRW;
    ra.hist_blocks = (used + 255) / 256;
    ra.part = (const double*)workspace;
    ra.init_acc = init_acc;
    hipLaunchKernelGGL(hist_reduce_kernel, dim3((unsigned)(ra.hist_blocks + (K + 63) / 64)),
                       dim3(64, RED_SLICES), 0, s, ra);
    DME_HIP_CHECK(hipGetLastError());
    if (sub > 1 && Nall > N * sub)
        return stats_dense_run<T>(X + N * sub * K, Nall - N * sub, K, ld, hist, nbins, histV, tbins,
                                  init_acc, status, overflow, workspace, workspace_bytes, stream, false);
    return DME_OK;
}

long long base_workspace_bytes(long long N, long long K) {
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    long long rounds = (N + (long long)cus * kRowsPerWgMax - 1) / ((long long)cus * kRowsPerWgMax);
    if (rounds < 1) rounds = 1;
    // per workgroup: K fp64 mean accumulators + one raw LDS histogram image
    return ((long long)cus * rounds * (K * 8 + kLdsBudgetBytes + 64) + 1024 + 255) & ~255LL;
}

template <typename T>
int stats_dense(const T* X, long long N, long long K, long long ld, u64* hist, long long nbins,
                u64* histV, long long tbins, double* init_acc, u64* status, u64* overflow, void* workspace,
                long long workspace_bytes, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K, "bad shape N=%lld K=%lld ld=%lld", N, K, ld);
    DME_REQUIRE(nbins >= 1 && tbins >= 1, "nbins/tbins must be >= 1");
    DME_REQUIRE(hist && histV && init_acc && status, "null output pointer");
    DME_REQUIRE(K * nbins < (1LL << 40), "hist too large");
    if (N == 0) return DME_OK;
    DME_REQUIRE(X != nullptr, "X is null");
    return stats_dense_run<T>(X, N, K, ld, hist, nbins, histV, tbins, init_acc, status, overflow,
                              workspace, workspace_bytes, stream, false);
}

}  // namespace

// The file is compiled twice (Makefile), once per element type, so that the two halves of the kernel
// instantiations build side by side: DME_STATS_PART 0 = uint16 (+ the shared entry), 1 = int32.
#ifndef DME_STATS_PART
#define DME_STATS_PART -1
#endif
#if DME_STATS_PART != 1
extern "C" int64_t dme_stats_workspace_bytes(int64_t N, int64_t K) {
    return (int64_t)base_workspace_bytes(N, (K + 7) & ~7LL);
}

extern "C" int dme_stats_dense_u16(const uint16_t* X, int64_t N, int64_t K, int64_t ld,
                                   uint64_t* hist, int64_t nbins, uint64_t* histV, int64_t tbins,
                                   double* init_acc, uint64_t* status, uint64_t* overflow,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    return stats_dense<unsigned short>(X, N, K, ld, (u64*)hist, nbins, (u64*)histV, tbins, init_acc,
                                       (u64*)status, (u64*)overflow, workspace, workspace_bytes, stream);
}

#endif
#if DME_STATS_PART != 0
extern "C" int dme_stats_dense_i32(const int32_t* X, int64_t N, int64_t K, int64_t ld,
                                   uint64_t* hist, int64_t nbins, uint64_t* histV, int64_t tbins,
                                   double* init_acc, uint64_t* status, uint64_t* overflow,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    return stats_dense<int>(X, N, K, ld, (u64*)hist, nbins, (u64*)histV, tbins, init_acc,
                            (u64*)status, (u64*)overflow, workspace, workspace_bytes, stream);
}
#endif
