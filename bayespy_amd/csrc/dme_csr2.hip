// dme_csr2.hip — statistics pass over CSR rows for vocabularies of up to 65 536 columns (the
// bag-of-words shape of BASELINE configs[3]: K = 50 000), second design.
//
// Reference semantics as in dme_csr.hip: CompressedRowData.appendRow over every row
// (DE/dme.py:113-123) and the CLI's mean accumulator (DE/fdp.py:87-89):
//   hist[k][v] += 1 per stored entry (k, v > 0),  histV[n_i] += 1,  init_acc[k] += v / (1 + n_i).
//
// The first design (dme_csr.hip: count pass, offsets, workgroup-wide radix batches behind six
// barriers each, 8-byte records) moved 4.5x the algorithmic bytes and ran at 8 % of the HBM peak.
// Here the entries cross memory twice, as 4-byte records, and no workgroup barrier sits on the path:
//
//   csr2_part_kernel    ONE WAVE OWNS A ROW: its lanes load the row's entries, the row total is a DPP
//                       wave reduction, and every entry becomes a 32-bit record
//                       {column-in-bucket : 10 | count : 8 | row total : 14} of its column bucket
//                       (1024 consecutive columns or fewer; at most 64 buckets).  Records are staged
//                       in wave-private LDS rings, one per bucket, and leave in aligned 64-byte units
//                       (16 records) to the wave's open chunk of that bucket; chunks (4 KiB) come
//                       from the workgroup's own range of chunk ids (sized from its rows' entries),
//                       so no counting pass and no global allocation counter are needed.
//                       Entries a record cannot hold (count >= 256 or row total >= 16384) go to
//                       global atomics.                                   [reads all, writes 4 B/entry]
//   csr2_sort_kernel    lists the chunks bucket by bucket (a counting sort of <= a few 1e5 ids)
//   csr2_bucket_kernel  the bucket-ordered chunk list is cut into equal pieces, one per workgroup; a
//                       piece counts its records in an LDS image of its bucket (32-bit bins
//                       1..BL per column, fp64 mean accumulators) and dumps the image of every
//                       (bucket, piece) run                                         [reads 4 B/entry]
//   csr2_reduce_kernel  adds the images of each bucket into hist / init_acc
//
// Integer results are exact and order-independent; init_acc is summed in arrival order (fp64), so
// its last bits vary from run to run, as on the other CSR paths.  A malformed rowptr (negative,
// decreasing, beyond nnz) is flagged in status[6]; such rows are not read, and rows that overlap
// can never overrun the workspace: the chunk pool refuses allocations beyond its size.
#include <atomic>

#include "dme_common.h"

namespace dme_csr2 {

constexpr int PBLOCK = 1024;           // partition kernel: 16 waves, one workgroup per CU
constexpr int PWAVES = PBLOCK / 64;
constexpr int MAXB = 64;               // buckets at most (one lane per bucket)
constexpr int UNIT = 16;               // records per flush unit (64 bytes)
constexpr int RING = 32;               // records a (wave, bucket) ring holds
constexpr int CHUNK = 1024;            // records per chunk (4 KiB)
constexpr int TBL = 2048;              // row totals below this are counted in LDS
constexpr int COLBITS = 10, VBITS = 8, TBITS = 14;
constexpr u32 VLIM = 1u << VBITS;      // counts a record can hold: 1 .. 255
constexpr u32 TLIM = 1u << TBITS;      // row totals a record can hold: 0 .. 16383
constexpr int BBLOCK = 1024;           // bucket / reduce kernels
constexpr int IMAGE_LDS_BYTES = 112 * 1024;
constexpr int WTAB = 2048;             // row totals whose weight 1 / (1 + n) is tabulated in LDS
constexpr int OVF = 2048;              // slots of an image's overflow table (counts beyond the bins)
#ifndef DME_CSR2_PIECES_PER_CU
#define DME_CSR2_PIECES_PER_CU 4
#endif

// the tile partition kernel (csr3_part_kernel below)
#ifndef DME_CSR3_T3
#define DME_CSR3_T3 512
#endif
#ifndef DME_CSR3_WGS_PER_CU
#define DME_CSR3_WGS_PER_CU (1024 / DME_CSR3_T3)
#endif
constexpr int T3 = DME_CSR3_T3;       // threads of a tile workgroup (at most 512)
constexpr int T3W = T3 / 64;
constexpr int EPT = 8;                // entries per thread
constexpr int TILE = T3 * EPT;        // entries per tile
constexpr int RW = T3;                // rows of a tile's window (one per thread)

struct Plan {
    int sh, CB, NB, BL, AC, G, W;
    int G3;                       // workgroups of the tile partition kernel
    long long chunks_per_wg3;     // chunk ids each of them owns
    long long off_flag, off_rstart;
    long long rows_per_wg, max_chunks;
    long long image_bytes;
    // workspace offsets (bytes)
    long long off_head, off_cbucket, off_cfill, off_sorted, off_bstart, off_bcount, off_bpos, off_images, off_chunks, bytes;
};

inline long long align256(long long x) { return (x + 255) & ~255LL; }

inline bool plan(long long N, long long nnz, long long K, long long nbins, Plan* p) {
    int sh = 8;
    while (((K + (1LL << sh) - 1) >> sh) > MAXB) ++sh;
    if (sh > COLBITS) return false;  // more than 64 buckets of 1024 columns: the first design
    p->sh = sh;
    p->CB = 1 << sh;
    p->NB = (int)((K + p->CB - 1) >> sh);
    int ac = 4;
    while (ac > 1 && 8LL * ac * p->CB > 16 * 1024) ac >>= 1;
    p->AC = ac;
    long long bl = (IMAGE_LDS_BYTES - 8LL * OVF - 8LL * ac * p->CB) / (4LL * p->CB);
    if (bl > nbins - 1) bl = nbins - 1;
    if (bl > (long long)VLIM - 1) bl = VLIM - 1;
    if (bl < 1) bl = 1;
    p->BL = (int)bl;
    p->image_bytes = 8LL * ac * p->CB + 4LL * p->CB * bl + 8LL * OVF;  // a multiple of 16
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    long long G = cus;
    if (G > (N + PWAVES - 1) / PWAVES) G = (N + PWAVES - 1) / PWAVES;
    if (G < 1) G = 1;
    p->G = (int)G;
    p->rows_per_wg = (N + G - 1) / G;
    p->W = cus * DME_CSR2_PIECES_PER_CU;
    // every record fills a slot; every (wave, bucket) pair may leave one chunk partly filled
    p->max_chunks = (nnz + CHUNK - 1) / CHUNK + G * ((long long)PWAVES * p->NB + 2) + 64;
    // the tile kernel: G3 workgroups with ceil(nnz / G3) entries each at most; every bucket's stream may
    // leave one chunk partly filled
    long long G3 = (nnz + 4LL * TILE - 1) / (4LL * TILE);
    if (G3 > (long long)cus * DME_CSR3_WGS_PER_CU) G3 = (long long)cus * DME_CSR3_WGS_PER_CU;
    if (G3 < 1) G3 = 1;
    p->G3 = (int)G3;
    const long long per3 = (nnz + G3 - 1) / G3;
    p->chunks_per_wg3 = (per3 + CHUNK - 1) / CHUNK + p->NB + 1;
    if (G3 * p->chunks_per_wg3 > p->max_chunks) p->max_chunks = G3 * p->chunks_per_wg3;
    if (p->max_chunks >= (1LL << 21)) return false;  // sorted entries hold 21-bit chunk ids
    long long o = 0;
    p->off_head = o;     o = align256(o + 256);
    p->off_flag = o;     o = align256(o + 256);
    p->off_bcount = o;   o = align256(o + 4 * MAXB);   // (flag, bcount and rstart lie side by side: one memset)
    p->off_rstart = o;   o = align256(o + 8 * (G3 + 1));
    p->off_cbucket = o;  o = align256(o + p->max_chunks);
    p->off_cfill = o;    o = align256(o + 2 * p->max_chunks);
    p->off_sorted = o;   o = align256(o + 4 * p->max_chunks);
    p->off_bstart = o;   o = align256(o + 4 * (MAXB + 2));
    p->off_bpos = o;     o = align256(o + 4 * MAXB);
    p->off_images = o;   o = align256(o + (long long)(p->NB + p->W) * p->image_bytes);
    p->off_chunks = o;   o = align256(o + p->max_chunks * CHUNK * 4LL + (long long)p->G3 * TILE * 4LL);  // + the tile kernel's dump areas
    p->bytes = o;
    return true;
}

struct PartArgs {
    const long long* rowptr;
    const int* colidx;
    const int* vals;
    long long N, nnz, K, nbins, tbins, rows_per_wg, max_chunks;
    int sh, NB, TB;  // TB = min(tbins, TBL)
    long long chunks_per_wg_extra;  // chunk ids a workgroup owns beyond its entries / CHUNK
    unsigned char* chunk_bucket;
    unsigned short* chunk_fill;
    u32* chunks;
    u64* hist;
    u64* histV;
    double* init_acc;
    u64* status;
    const u32* rowptr_bad;  // set by csr3_check_kernel: the row-wise kernel below runs only then
    u64* overflow;          // store for counts >= nbins (dme_common.h), or NULL
};

// Per-wave staging in LDS.  All of it is private to one wave: LDS serves a wave's accesses in
// program order, so no barrier is needed between the phases below, only compiler ordering points.
//
// cnt[b] only ever grows: record number p of the (wave, bucket) stream sits in ring slot p mod RING,
// belongs to unit p / UNIT (16 records = one aligned 64-byte store unit) and to the wave's chunk
// number p / CHUNK of that bucket, at offset p mod CHUNK.  The lane that stages the LAST record of a
// unit writes the unit out on the spot (four 16-byte stores from its own lane): no scan over the
// buckets, no ring bookkeeping.  A step stages at most two units' worth per bucket at a time, so a
// unit is never overwritten before it has left (see `process`).
struct WaveStage {
    u32 ring[MAXB][RING];  // RING = 2 units
    u32 cnt[MAXB];         // records staged so far (monotone)
    u32 cptr[MAXB][2];     // chunk ids of the bucket's chunk numbers of even / odd parity
};

// Compiler-only ordering point (no instruction, in particular no s_waitcnt: a hardware fence here
// made every staging phase wait for the wave's outstanding global stores and prefetched loads).
__device__ __forceinline__ void wave_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Chunk number `seq` of (wave, bucket b) begins: take an id from the workgroup's range.  Called by one
// lane.  Every chunk is recorded as full; the last one of a stream is corrected by the final flush.
__device__ __forceinline__ u32 open_chunk(WaveStage& st, const PartArgs& a, int b, u32 seq,
                                          u32* wg_next, long long cbase, long long climit) {
    // an LDS counter, no global atomic (one pool head for the whole chip saturates at ~90
    // allocations per microsecond: 4.4 ms for the 385 k chunks of a 4 M-row shard)
    const long long cl = cbase + (long long)atomicAdd(wg_next, 1u);
    u32 id = 0xFFFFFFFFu;
    if (cl < climit) {
        id = (u32)cl;
        a.chunk_bucket[id] = (unsigned char)b;
        a.chunk_fill[id] = (unsigned short)CHUNK;
    } else {
        atomicAdd(&a.status[6], 1ULL);  // only overlapping rows of a malformed rowptr get here
    }
    st.cptr[b][seq & 1] = id;
    return id;
}

__global__ __launch_bounds__(PBLOCK) void csr2_part_kernel(PartArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    WaveStage* stages = (WaveStage*)lds_raw;
    u32* tb = (u32*)(lds_raw + sizeof(WaveStage) * PWAVES);  // [TB] row-total bins
    __shared__ u64 sh_stat[8];
    __shared__ u32 wg_next;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.rowptr_bad && *a.rowptr_bad == 0) return;  // a well-formed rowptr: csr3_part_kernel has done the work
    WaveStage& st = stages[wave];
    for (int i = tid; i < a.TB; i += PBLOCK) tb[i] = 0;
    if (tid < 8) sh_stat[tid] = 0;
    if (tid == 0) wg_next = 0;
    st.cnt[lane] = 0;
    st.cptr[lane][0] = 0xFFFFFFFFu;
    st.cptr[lane][1] = 0xFFFFFFFFu;
    __syncthreads();

    const long long R0 = (long long)blockIdx.x * a.rows_per_wg;
    long long R1 = R0 + a.rows_per_wg;
    if (R1 > a.N) R1 = a.N;
    const u32 cmask = (1u << a.sh) - 1;
    const u32 vmax = a.nbins < (long long)VLIM ? (u32)a.nbins : VLIM;  // counts a record and hist hold
    const u32 tmax = (u32)a.TB < TLIM ? (u32)a.TB : TLIM;              // totals the LDS bins and a record hold
    u32 maxv = 0;
    u64 maxt = 0, bad = 0, big = 0, nrows = 0, badrows = 0;
    bool pool_ok = true;
    // This workgroup's chunk ids: [cbase, climit).  Its rows hold E = rowptr[R1] - rowptr[R0] entries,
    // so it needs at most ceil(E / CHUNK) full chunks plus one partly filled chunk per (wave,
    // bucket); ranges of successive workgroups cannot overlap because
    // floor((S + E) / CHUNK) - floor(S / CHUNK) >= ceil(E / CHUNK) - 1.  (A malformed rowptr only
    // shrinks a range: ids are clamped to the pool, and a range that runs out flags status[6].)
    long long cbase = 0, climit = 0;
    if (R0 < R1) {
        const long long s0 = a.rowptr[R0] - a.rowptr[0], s1 = a.rowptr[R1] - a.rowptr[0];
        cbase = s0 / CHUNK + (long long)blockIdx.x * a.chunks_per_wg_extra;
        climit = cbase + (s1 > s0 ? (s1 - s0) / CHUNK : 0) + a.chunks_per_wg_extra;
        if (cbase < 0) cbase = 0;
        if (climit > a.max_chunks) climit = a.max_chunks;
        if ((long long)(blockIdx.x + 1) < (long long)gridDim.x) {
            // never beyond the next workgroup's base
            const long long nb0 = (a.rowptr[R1] - a.rowptr[0]) / CHUNK + (long long)(blockIdx.x + 1) * a.chunks_per_wg_extra;
            if (climit > nb0) climit = nb0;
        }
    }

    // Stages the records of the lanes with `ok` set (count in 1 .. VLIM - 1 and below nbins, column
    // below K, row total below TLIM) and writes out the units they complete.
    auto stage = [&](int c, u32 v, u32 total, bool ok) {
        const int b = ok ? (c >> a.sh) : 0;
        const u32 rec = ((u32)c & cmask) | (v << COLBITS) | (total << (COLBITS + VBITS));
        if (!__any(ok)) return;
        // every record gets its number in its bucket's stream at once; pos0 = the stream's length
        // before this step (the same for all lanes of a bucket)
        u32 pos0 = 0, pos = 0;
        if (ok) {
            pos0 = st.cnt[b];
            wave_fence();
            pos = atomicAdd(&st.cnt[b], 1u);
        }
        wave_fence();
        // Phase q stages the records of units first + 2q, first + 2q + 1 (first = the unit the
        // stream stood in): with the ring two units long, a unit is written out before a record two
        // units later lands on its slots.  One phase unless a bucket takes more than 16 of the
        // step's records.
        const u32 rel = (pos >> 4) - (pos0 >> 4);
        bool pending = ok;
        for (u32 q = 0; __any(pending); ++q) {
            const bool now = pending && (rel >> 1) == q;
            if (now) st.ring[b][pos & (RING - 1)] = rec;
            wave_fence();
            const bool owner = now && (pos & (UNIT - 1)) == (u32)(UNIT - 1);  // staged a unit's last record
            const u32 seq = pos >> 10;                         // chunk number of the stream (CHUNK = 2^10)
            const u32 off = (pos & (CHUNK - 1)) - (UNIT - 1);  // the unit's offset in its chunk
            // units that begin a chunk open it first; only then does any lane look a chunk id up
            // (two units of one bucket can leave in the same phase, the second of them mid-chunk)
            if (owner && off == 0) open_chunk(st, a, b, seq, &wg_next, cbase, climit);
            wave_fence();
            if (owner) {
                const u32 id = st.cptr[b][seq & 1];
                if (id != 0xFFFFFFFFu) {
                    const uint4* src = (const uint4*)&st.ring[b][pos & (RING - 1) & ~(UNIT - 1)];
                    uint4* dst = (uint4*)(a.chunks + (long long)id * CHUNK + off);
                    const uint4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
                    dst[0] = x0;
                    dst[1] = x1;
                    dst[2] = x2;
                    dst[3] = x3;
                }
            }
            wave_fence();
            if (now) pending = false;
        }
    };

    // one entry per lane: classify, stage the record, flush full units
    auto process = [&](int c, int v, u64 total, bool active) {
        bool ok = active && v != 0;
        if (ok && (v < 0 || (u32)c >= (u32)a.K)) {
            bad++;
            ok = false;
        }
        if (ok) maxv = (u32)v > maxv ? (u32)v : maxv;
        if (ok && (long long)v >= a.nbins) {  // beyond the caller's histogram: to the overflow store, else reported
            if (!dme_overflow_put(a.overflow, (u64)c, (u64)v, 1ULL)) big++;  // (the mean accumulator takes it anyway)
            const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
            atomicAdd(&a.init_acc[c], __dmul_rn((double)v, w));
            ok = false;
        }
        if (ok && ((u32)v >= VLIM || total >= (u64)TLIM)) {  // a record cannot hold it
            atomicAdd(&a.hist[(long long)c * a.nbins + v], 1ULL);
            const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
            atomicAdd(&a.init_acc[c], __dmul_rn((double)v, w));
            ok = false;
        }
        stage(c, (u32)v, (u32)total, ok);
    };

    // Software pipeline over the wave's rows: the entries of the next row and the pointers of the next
    // 64 rows are in flight while the current row is staged.
    auto row_ok = [&](long long p0, long long p1) { return p0 >= 0 && p1 >= p0 && p1 <= a.nnz; };
    auto load_entries = [&](long long p0, long long p1, int& c, int& v) {
        c = 0;
        v = 0;
        if (row_ok(p0, p1) && p1 - p0 <= 64 && lane < p1 - p0) {
            c = __builtin_nontemporal_load(a.colidx + p0 + lane);
            v = __builtin_nontemporal_load(a.vals + p0 + lane);
        }
    };
    // Row pointers come 64 rows at a time as VECTOR loads (lane l holds the pointers of the wave's
    // row 64 j + l) and are handed out by v_readlane.  Scalar loads of them would share the LDS
    // operations' counter (lgkmcnt, no in-order guarantee): every wait for an LDS result would then
    // also wait out the memory latency of the pointer prefetch, once per row.
    const long long nrw = R1 > R0 + wave ? (R1 - R0 - wave + PWAVES - 1) / PWAVES : 0;  // this wave's rows
    auto load_batch = [&](long long j, long long& P0, long long& P1) {
        const long long i = j * 64 + lane;
        P0 = 0;
        P1 = -1;
        if (i < nrw) {
            const long long r = R0 + wave + i * PWAVES;
            P0 = a.rowptr[r];
            P1 = a.rowptr[r + 1];
        }
    };
    auto lane_val = [&](long long P, int idx) -> long long {
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)P, idx);
        const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)P >> 32), idx);
        return (long long)((u64)lo | ((u64)hi << 32));
    };
    long long C0, C1, N0, N1;
    load_batch(0, C0, C1);
    load_batch(1, N0, N1);
    long long pA0 = 0, pA1 = -1;
    int cN = 0, vN = 0;
    if (nrw > 0) {
        pA0 = lane_val(C0, 0);
        pA1 = lane_val(C1, 0);
        load_entries(pA0, pA1, cN, vN);
    }
    for (long long i = 0; i < nrw; ++i) {
        const long long p0 = pA0, p1 = pA1;
        const int c0 = cN, v0 = vN;
        const long long ni = i + 1;
        if ((ni & 63) == 0) {  // wave-uniform: the next batch of pointers becomes current
            C0 = N0;
            C1 = N1;
            load_batch((ni >> 6) + 1, N0, N1);
        }
        if (ni < nrw) {
            const int idx = __builtin_amdgcn_readfirstlane((int)(ni & 63));
            pA0 = lane_val(C0, idx);
            pA1 = lane_val(C1, idx);
            load_entries(pA0, pA1, cN, vN);
        }
        // ---- the plain row (nearly all of them): at most 64 entries, every count in 0 .. vmax - 1, every
        //      column in range, a total the LDS bins and a record hold: one straight path ---------------
        if (row_ok(p0, p1) && p1 - p0 <= 64) {
            const bool in = lane < (int)(p1 - p0);
            const u32 uv = (u32)v0;  // a negative count reads as a huge one
            if (!__any(in && (uv >= vmax || (u32)c0 >= (u32)a.K))) {
                const u32 tot = dme_wave_sum_u32(uv);  // lanes beyond the row hold 0; 64 * 255 cannot wrap
                if (tot < tmax) {
                    nrows++;
                    maxv = uv > maxv ? uv : maxv;
                    if (tot > 0) {
                        maxt = (u64)tot > maxt ? (u64)tot : maxt;
                        if (lane == 0) atomicAdd(&tb[tot], 1u);
                    }
                    stage(c0, uv, tot, in && uv != 0);
                    continue;
                }
            }
        }
        // ---- every other row: the general path ----------------------------------------------------------
        if (!row_ok(p0, p1)) {  // malformed: the row is not read (wave-uniform)
            badrows++;
            continue;
        }
        nrows++;
        const long long len = p1 - p0;
        // ---- the row total (DE/dme.py:121 sum(row)) ------------------------------------------
        u64 total = 0;
        if (len <= 64) {
            const u32 pos = v0 > 0 ? (u32)v0 : 0u;
            if (__builtin_expect(!__any(pos >= (1u << 24)), 1)) total = dme_wave_sum_u32(pos);
            else total = dme_wave_sum_u64((u64)pos);
        } else {
            u64 s = 0;
            for (long long q = p0 + lane; q < p1; q += 64) {
                const int v = a.vals[q];
                s += (u64)(v > 0 ? v : 0);
            }
            total = dme_wave_sum_u64(s);
        }
        if (total > 0) {
            maxt = total > maxt ? total : maxt;
            if (lane == 0) {
                if (total < (u64)a.TB) atomicAdd(&tb[total], 1u);
                else if ((long long)total < a.tbins) atomicAdd(&a.histV[total], 1ULL);
                else atomicAdd(&sh_stat[2], 1ULL);
            }
        }
        // ---- the entries -----------------------------------------------------------------------
        if (len <= 64) {
            process(c0, v0, total, lane < len);
        } else {
            for (long long q0 = p0; q0 < p1; q0 += 64) {
                const long long q = q0 + lane;
                const bool in = q < p1;
                const int c = in ? a.colidx[q] : 0;
                const int v = in ? a.vals[q] : 0;
                process(c, v, total, in);
            }
        }
    }
    // ---- final flush: lane b writes the last, partial unit of bucket b's stream and its chunk's fill ---
    wave_fence();
    if (lane < a.NB && pool_ok) {
        const u32 n = st.cnt[lane];
        if (n > 0) {
            const u32 part = n & (UNIT - 1);
            const u32 seq = (n - 1) >> 10;            // the stream's last chunk
            u32 id;
            if (part && (n & (CHUNK - 1)) == part)    // the partial unit opens a chunk of its own
                id = open_chunk(st, a, lane, seq, &wg_next, cbase, climit);
            else
                id = st.cptr[lane][seq & 1];
            if (id != 0xFFFFFFFFu) {
                const u32 first = n - part;           // stream number of the unit's first record
                for (u32 j = 0; j < part; ++j)
                    a.chunks[(long long)id * CHUNK + ((first + j) & (CHUNK - 1))] =
                        st.ring[lane][(first + j) & (RING - 1)];
                a.chunk_fill[id] = (unsigned short)(n - (seq << 10));
            }
        }
    }
    // ---- statistics of the pass ------------------------------------------------------------------------
    if (lane == 0) {
        if (nrows) atomicAdd(&sh_stat[0], nrows);
        if (badrows) atomicAdd(&sh_stat[6], badrows);
    }
    bad = dme_wave_sum_u64(bad);
    big = dme_wave_sum_u64(big);
    const u32 mv = dme_wave_max_u32(maxv);
    if (lane == 0) {
        if (bad) atomicAdd(&sh_stat[3], bad);
        if (big) atomicAdd(&sh_stat[1], big);
        if (mv) atomicMax(&sh_stat[4], (u64)mv);
        if (maxt) atomicMax(&sh_stat[5], maxt);
    }
    __syncthreads();
    for (int i = tid; i < a.TB; i += PBLOCK) {
        const u32 cn = tb[i];
        if (cn) atomicAdd(&a.histV[i], (u64)cn);
    }
    if (tid <= 3 && sh_stat[tid]) atomicAdd(&a.status[tid], sh_stat[tid]);
    if (tid >= 4 && tid <= 5 && sh_stat[tid]) atomicMax(&a.status[tid], sh_stat[tid]);
    if (tid == 6 && sh_stat[6]) atomicAdd(&a.status[6], sh_stat[6]);
}


// =====================================================================================================
// Tile partition kernel (round 3).  The row-wise kernel above keeps one wave on one row: 27 % of the
// lanes idle at ~47 entries per row, five dependent LDS round trips per row, one row of loads in flight
// (4.85 ms for the 12.5 M-row shard of BASELINE configs[3]; SQ_WAIT_ANY 63 % of the wave cycles).  Here a
// 512-thread workgroup takes the entries themselves, TILE = 4096 consecutive entries at a time, eight per
// thread whatever the row lengths:
//   P1  thread j looks at row j of the tile's row window (512 row pointers): which rows start in the
//       tile, which row straddles its end; the eight counts of every thread go into a prefix sum
//   P2  prefix sums to LDS (PX); every row writes its window index at the position where it starts (F);
//       every entry takes its number in its bucket's run with one returning LDS add; waves 0 / 1 add up
//       what the first / last row of the tile holds outside the tile (read from L2, the neighbour tile
//       has just loaded or is about to load it)
//   P3  row totals = differences of PX (+ the outside parts): histV, maxima, row count; lane b of wave 0
//       turns the bucket counts into offsets and takes chunk ids for bucket b's run; every entry finds
//       its row: the largest F at or before it (running maximum in the thread, DPP max-scan in the wave)
//   P4  records {column-in-bucket | count | row total} go to their place in the bucket-ordered staging
//       area; what a record cannot hold goes to global atomics, as in the row-wise kernel
//   P5  wave w writes the runs of buckets w, w + 8, ... to the buckets' streams: consecutive lanes,
//       consecutive records (a run of a 4096-entry tile is ~84 records at 49 buckets)
// Four barriers per tile; the loads of the next tile (entries and row pointers) are issued in P2.
// Rows are OWNED (counted, added to histV) by the tile their index falls to: tile boundaries advance in
// entries (<= TILE) and in rows (<= 512), whichever comes first, so empty rows and one-entry rows cannot
// stall it.  Workgroup g owns the entries [E0 + g S, E0 + (g + 1) S) of the matrix, E0 = rowptr[0],
// S = ceil((rowptr[N] - E0) / G): its chunk ids are a fixed range of ceil(S / CHUNK) + NB + 1, and the
// row that holds its first entry comes from csr3_check_kernel, which also proves rowptr well-formed
// (non-negative, non-decreasing, within nnz).  A rowptr that is not leaves everything to the row-wise
// kernel: rows that overlap or run backwards have no place in a partition of the entries.
// =====================================================================================================
// -DDME_CSR3_TS: thread 0 of every workgroup adds the shader cycles (s_memtime) it spends in each phase
// of the tile loop to the first words of the workspace (benchmarks/csr_phases.py reads them).
#ifdef DME_CSR3_TS
#define TS_DECL u64 ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 ts_last = __builtin_amdgcn_s_memtime();
#define TS_MARK(i) { const u64 ts_now = __builtin_amdgcn_s_memtime(); ts_acc[i] += ts_now - ts_last; ts_last = ts_now; }
#define TS_FLUSH(p) if (threadIdx.x == 0) { for (int i = 0; i < 8; ++i) atomicAdd((u64*)(p) + i, ts_acc[i]); }
#else
#define TS_DECL
#define TS_MARK(i)
#define TS_FLUSH(p)
#endif

constexpr u32 HUGE_V = 1u << 19;      // counts from here on: row totals are added up in 64 bits, row by row

struct TileArgs {
    PartArgs a;
    u32* flag;             // written by the check kernel: != 0 = malformed rowptr
    u64* ts;               // phase time stamps (-DDME_CSR3_TS builds)
    long long* rstart;     // [G + 1] number of rows that start before E0 + g S; [G]: before rowptr[N]
    u32* bcount;           // [MAXB] chunks that hold records, per bucket (zeroed per call): what the chunk sort starts from
    int G;
    long long chunks_per_wg;
};

__global__ __launch_bounds__(256) void csr3_check_kernel(const long long* rowptr, long long N, long long nnz,
                                                         int G, u32* flag, long long* rstart) {
    const long long E0 = rowptr[0], E1 = rowptr[N];
    long long span = E1 - E0;
    if (span < 0) span = 0;
    long long S = (span + G - 1) / G;
    if (S < 1) S = 1;
    bool bad = false;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += stride) {
        const long long p0 = rowptr[j], p1 = rowptr[j + 1];
        if (p0 < 0 || p1 < p0 || p1 > nnz) {
            bad = true;
        } else if (p1 > p0 && p0 >= E0 && p1 <= E1) {
            // the boundaries E0 + g S in (p0, p1]: row j is the last one that starts before them
            // (a 32-bit division whenever the span allows, a third of a 64-bit one's cost; the second one
            // only for the few rows that do hold a boundary)
            long long g_lo, g_hi;
            if (span < (1LL << 32)) {
                const u32 x0 = (u32)(p0 - E0), x1 = (u32)(p1 - E0), q1 = x1 / (u32)S;
                g_hi = (long long)q1;
                g_lo = (long long)((u64)x0 >= (u64)q1 * (u64)S ? q1 : x0 / (u32)S) + 1;
            } else {
                g_lo = (p0 - E0) / S + 1;
                g_hi = (p1 - E0) / S;
            }
            if (g_hi > G - 1) g_hi = G - 1;
            for (long long g = g_lo; g <= g_hi; ++g) rstart[g] = j + 1;
            if (p1 == E1) rstart[G] = j + 1;
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__device__ __forceinline__ u32 wave_incl_scan_max_u32(u32 v) {
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
}
// the value of the lane below (lane 0: 0)
__device__ __forceinline__ u32 wave_shr1_u32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}

typedef u32 u32x4v __attribute__((ext_vector_type(4)));
typedef u32x4v u32x4v_u __attribute__((aligned(4)));

struct TileLds {
    u32 F[TILE];          // window index + 1 of the row that starts at an entry position (0: none)
    u32 PX[TILE + 4];     // exclusive prefix sums of the tile's counts; from P4 on: the staged records
    u64 RT[RW];           // row totals of the window's rows
    u32 RT32[RW];         // ... saturated to 32 bits (what the plain path reads)
    u32 cnt[MAXB + 4];    // records of the tile per bucket (P2), zeroed again in P3; [MAXB]: sink
    // per bucket {first staged position of its run, records that still fit the stream's open chunk,
    // destination of the run's record 0 in that chunk, destination of record 0 in the chunks taken for
    // this tile (their ids are consecutive)}: destination of record i = i < .y ? .z + i : .w + i
    u32x4v prm[MAXB + 1];
    unsigned char SB[TILE];  // bucket of every staged record (8-byte aligned: reset eight slots at a time)
    u32 staged;           // records of the tile in the staging area
    u32 len[MAXB];        // records written to the bucket's stream so far
    u32 cur[MAXB];        // id of the stream's last chunk
    u32 nch[MAXB];        // chunks the stream has taken
    u32 wsum[8], wrows[8];   // per wave (T3W <= 8; the rest stays 0)
    u32 WR[T3W];          // window index + 1 of the row that holds the entry in front of wave w's first
    u32 straddle;         // window index + 1 of the row that runs over the tile's end
    u32 wg_next;          // chunk ids taken from the workgroup's range
    u32 huge;             // a count >= HUGE_V in the tile
    u32 cut;              // the tile ends here (relative): the first entry of the row that would run over its end
    u64 head_extra, tail_extra;
    u64 stat[8];
};

// An entry the plain path does not take: a count the record (>= 256) or the caller's histogram (>= nbins)
// cannot hold, a row total a record cannot hold, a negative count, a column out of range.  Same rules as
// the row-wise kernel's `process`.  Out of line: its registers and the division stay out of the tile loop.
// Returns 1 for a bad entry (status[3]), 2 for a count beyond the caller's histogram (status[1]), else 0.
__device__ __noinline__ int csr3_rare_entry(u64* hist, double* init_acc, long long nbins, long long K, u32 c,
                                            u32 v, u64 total, u64* overflow) {
    if ((int)v < 0 || c >= (u32)K) return 1;
    int r = 0;
    if ((long long)v < nbins) atomicAdd(&hist[(long long)c * nbins + v], 1ULL);
    else if (!dme_overflow_put(overflow, (u64)c, (u64)v, 1ULL)) r = 2;  // reported, not counted; the mean accumulator still takes it
    const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
    atomicAdd(&init_acc[c], __dmul_rn((double)v, w));
    return r;
}

__device__ __forceinline__ u32 uni(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }

__global__ __launch_bounds__(T3, 4) void csr3_part_kernel(TileArgs ta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const PartArgs& a = ta.a;
    if (*ta.flag != 0) return;  // malformed rowptr: the row-wise kernel takes the call
    TileLds& L = *(TileLds*)lds_raw;
    u32* tb = (u32*)(lds_raw + sizeof(TileLds));  // [TB] row-total bins
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = ta.G, g = blockIdx.x;
    const long long N = a.N;
    // Row pointers are below 2^32 on this path (nnz < 2^32 and the check kernel has seen every pointer inside
    // [0, nnz]): their low words are read, and the tile loop's positions are 32-bit scalars.
    const u32* rp = (const u32*)a.rowptr;
    const u32 E0 = rp[0], E1 = rp[2 * N];
    u32 S = (u32)(((u64)(E1 - E0) + (u64)G - 1) / (u64)G);
    if (S < 1) S = 1;
    const u64 A64 = (u64)E0 + (u64)g * S;
    const u32 A = A64 > (u64)E1 ? E1 : (u32)A64;
    const u64 Ae64 = (u64)A + S;
    const u32 Aend = Ae64 > (u64)E1 ? E1 : (u32)Ae64;
    const bool last = g == G - 1;
    if (A >= E1 && !last) return;  // nothing of the matrix falls to this workgroup

    for (int i = tid; i < TILE; i += T3) L.F[i] = 0;
    for (int i = tid; i < a.TB; i += T3) tb[i] = 0;
    if (tid < MAXB) {
        L.cnt[tid] = 0;
        L.len[tid] = 0;
        L.cur[tid] = 0xFFFFFFFFu;
        L.nch[tid] = 0;
    }
    if (tid < 8) L.stat[tid] = 0;
    if (tid < T3W) L.WR[tid] = 0;
    if (tid < 8) {
        L.wsum[tid] = 0;
        L.wrows[tid] = 0;
    }
    if (tid == 0) {
        L.straddle = 0;
        L.cut = 0xFFFFFFFFu;
        L.wg_next = 0;
        L.huge = 0;
        L.head_extra = 0;
        L.tail_extra = 0;
    }
    const long long cbase = (long long)g * ta.chunks_per_wg;
    long long climit = cbase + ta.chunks_per_wg;
    if (climit > a.max_chunks) climit = a.max_chunks;

    long long jown = A >= E1 ? ta.rstart[G] : ta.rstart[g];  // first row this workgroup owns
    if (jown > N) jown = N;
    long long jb = jown;                                       // first row of the window
    if (A < E1 && jown >= 1 && rp[2 * jown] > A) jb = jown - 1;  // the row that holds entry A began earlier
    u32 apos = A;

    const u32 cmask = (1u << a.sh) - 1;
    const u32 vmax = a.nbins < (long long)VLIM ? (u32)a.nbins : VLIM;  // counts a record and hist hold
    // statistics of the pass: the two maxima in registers, the rare events straight to LDS (L.stat)
    u32 maxv = 0, maxt32 = 0;
    const long long jown_first = jown;

    // ---- loads of a tile: three row pointers and eight entries per thread -------------------------------
    auto load_window = [&](long long jb_, u32& q0, u32& q1, u32& qrw) {
        const long long rem = N - jb_;  // rows from the window's first one to the matrix' end (>= 0)
        const u32 rem32 = rem > (long long)RW ? (u32)RW : (u32)rem;
        const u32* base = rp + 2 * jb_;
        const u32 j0 = (u32)tid < rem32 ? (u32)tid : rem32, j1 = (u32)tid + 1 < rem32 ? (u32)tid + 1 : rem32;
        q0 = base[2 * j0];
        q1 = base[2 * j1];
        qrw = base[2 * rem32];
    };
    auto load_entries = [&](u32 ap, u32 (&c)[EPT], u32 (&v)[EPT]) {
        const u32 left = E1 - ap;  // entries from the tile's first one to the matrix' end
        const int* cb = a.colidx + ap;
        const int* vb = a.vals + ap;
#pragma unroll
        for (int h = 0; h < EPT / 4; ++h) {
            const u32 eh = (u32)tid * EPT + 4 * h;
            if (eh + 3 < left) {
                const u32x4v cc = __builtin_nontemporal_load((const u32x4v_u*)(cb + eh));
                const u32x4v vv = __builtin_nontemporal_load((const u32x4v_u*)(vb + eh));
                c[4 * h] = cc.x; c[4 * h + 1] = cc.y; c[4 * h + 2] = cc.z; c[4 * h + 3] = cc.w;
                v[4 * h] = vv.x; v[4 * h + 1] = vv.y; v[4 * h + 2] = vv.z; v[4 * h + 3] = vv.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = eh + i < left;
                    c[4 * h + i] = in ? (u32)cb[eh + i] : 0u;
                    v[4 * h + i] = in ? (u32)vb[eh + i] : 0u;
                }
            }
        }
    };
    // sum of the positive counts in [lo, hi) by one wave (what a row holds outside the tile)
    auto wave_row_part = [&](u32 lo, u32 hi) -> u64 {
        u64 s = 0;
        const u64 n = hi > lo ? (u64)(hi - lo) : 0;
        for (u64 o = (u64)lane; o < n; o += 64) {
            const int x = a.vals[(u64)lo + o];
            s += (u64)(x > 0 ? x : 0);
        }
        return dme_wave_sum_u64(s);
    };

    // staging slots nobody filled are stored too (always eight stores per thread: the wait for the next
    // tile's loads, issued before them, can then count them out): to this workgroup's dump area behind the pool
    const u32 dump = (u32)(a.max_chunks << 10) + (u32)g * TILE;
    u32 nq0, nq1, nqrw;
    u32 nc[EPT], nv[EPT];
    load_window(jb, nq0, nq1, nqrw);
    load_entries(apos, nc, nv);
    __syncthreads();

    TS_DECL
    while (true) {
        TS_MARK(5)
        const u32 q0 = nq0, q1 = nq1, qrw = uni(nqrw);
        u32 c[EPT], v[EPT];
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            c[i] = nc[i];
            v[i] = nv[i];
        }
        // ---- the tile: entries [apos, apos + n_ent), rows [jown, jownN) -------------------------------
        // First the tile as far as it could reach: TILE entries, the workgroup's end or the window's end.
        // A row that starts inside it and runs over that end is left to the next tile whole (the tile is
        // CUT at its first entry; agreed behind barrier B1): a row's entries then lie in one tile, and its
        // total is a difference of two prefix sums.  Only a row that began before the tile (longer than a
        // tile, or cut by the workgroup's range) needs what it holds outside, read again from memory.
        const u32 left = Aend - apos;                       // entries left of the workgroup's range
        const u32 lim = left < (u32)TILE ? left : (u32)TILE;
        const long long rows_left = N - jb;
        const u32 wrows = rows_left > (long long)RW ? (u32)RW : (u32)rows_left;  // rows the window holds
        const u32 qrw_rel = qrw - apos;                     // the window's end is never in front of the tile
        const bool exhausted = rows_left > (long long)RW && qrw_rel < lim;  // its rows end before the entries do
        const int n0 = (int)(exhausted ? qrw_rel : lim);
        const int jrel = (int)(jown - jb);                  // 1: window row 0 began before the tile
        const bool vrow = (u32)tid < wrows;
        // the row's first entry and its end relative to the tile, clamped to [-1, TILE + 1]
        const long long sl = (long long)q0 - (long long)apos, tl = (long long)q1 - (long long)apos;
        const int s0 = sl < 0 ? -1 : (sl > TILE ? TILE + 1 : (int)sl);
        const int t0 = tl < 0 ? -1 : (tl > TILE ? TILE + 1 : (int)tl);
        const bool over_end = vrow && s0 < n0 && t0 > n0;  // at most one row
        const bool cut_here = over_end && s0 > 0;
        const bool owned = vrow && tid >= jrel && !cut_here && (exhausted || s0 < n0 || (last && apos + (u32)n0 >= E1));

        // ---- P1 ------------------------------------------------------------------------------------------
        // (entries behind the tile's end are summed too: the sums behind it are not read; a count of 2^19 or
        // more anywhere sends the tile's row totals down the 64-bit path, so the 32-bit sums may then wrap)
        u32 vp[EPT];
        u32 tsum = 0, orv = 0;
        const int k0 = tid * EPT;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int x = (int)v[i];
            const u32 pv = x > 0 ? (u32)x : 0u;
            orv |= pv;
            vp[i] = tsum;  // exclusive inside the thread
            tsum += pv;
        }
        const bool hg = orv >= HUGE_V;
        const u32 incl = dme_wave_incl_scan_u32(tsum);
        if (lane == 63) L.wsum[wave] = incl;
        {
            const u64 ob = __ballot(owned);
            if (lane == 0) L.wrows[wave] = (u32)__popcll(ob);
            if (__any(hg) && lane == 0) L.huge = 1;
        }
        if (over_end) {
            L.straddle = (u32)tid + 1;
            if (cut_here) L.cut = (u32)s0;
        }
        TS_MARK(0)
        __syncthreads();  // B1
        TS_MARK(6)

        // ---- P2 ------------------------------------------------------------------------------------------
        u32 rows_owned = 0, wbase = 0;
        {
            const u32x4v r0 = *(const u32x4v*)&L.wrows[0], r1 = *(const u32x4v*)&L.wrows[4];
            const u32x4v x0 = *(const u32x4v*)&L.wsum[0], x1 = *(const u32x4v*)&L.wsum[4];
            const u32 r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const u32 x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int w = 0; w < T3W; ++w) {
                rows_owned += uni(r[w]);
                wbase += w < wave ? uni(x[w]) : 0u;
            }
        }
        const u32 strad = uni(L.straddle), cutv = uni(L.cut);
        const bool has_huge = uni(L.huge) != 0;
        const bool is_cut = cutv != 0xFFFFFFFFu;
        const int n_ent = is_cut ? (int)cutv : n0;
        const u32 anext = apos + (u32)n_ent;
        const bool tail = strad != 0 && !is_cut;           // a row that began at or before apos runs over the end
        const int s_rel = s0 > n_ent ? n_ent + 1 : s0, t_rel = t0 > n_ent ? n_ent + 1 : t0;
        const long long jownN = jown + rows_owned;
        const long long jbN = strad ? jb + (long long)strad - 1 : jownN;
        const bool cont = anext < Aend || (last && jownN < N);
        if (cont) {
            load_window(jbN, nq0, nq1, nqrw);
            if (anext < Aend) load_entries(anext, nc, nv);
        }
        {
            const u32 tbase = wbase + incl - tsum;  // entries before this thread
            u32x4v* px = (u32x4v*)&L.PX[k0];
            px[0] = u32x4v{tbase + vp[0], tbase + vp[1], tbase + vp[2], tbase + vp[3]};
            px[1] = u32x4v{tbase + vp[4], tbase + vp[5], tbase + vp[6], tbase + vp[7]};
            if (tid == T3 - 1) L.PX[TILE] = tbase + tsum;
        }
        if (vrow && s_rel >= 0 && s_rel < n_ent) atomicMax(&L.F[s_rel], (u32)tid + 1);
        if (vrow && t_rel > s_rel && s_rel < n_ent) {
            // this row holds the entry in front of a wave's first one (position 512 w - 1): that wave's
            // entries start out in this row
            for (int w = s_rel < 0 ? 1 : (s_rel >> 9) + 1; w < T3W && (w << 9) <= t_rel; ++w) L.WR[w] = (u32)tid + 1;
        }
        u32 okm = 0;  // bit i: entry i goes to the staging area; bit 31: an entry that does not (and is not a zero)
        {
            // eight returning adds side by side, one wait (entries that are not staged add to a sink word:
            // a branch around each add made every one of them wait for its own result)
            u32 rk[EPT];
            u32 nonok = 0;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                v[i] = k0 + i < n_ent ? v[i] : 0u;  // entries behind the tile's end: zeros from here on
                const bool ok = v[i] - 1u < vmax - 1u && c[i] < (u32)a.K;  // count in 1 .. vmax - 1, column in range
                okm |= ok ? 1u << i : 0u;
                const u32 vm = ok ? v[i] : 0u;
                maxv = vm > maxv ? vm : maxv;
                nonok |= v[i] ^ vm;  // != 0: a count that is there and is not staged
                rk[i] = atomicAdd(&L.cnt[ok ? c[i] >> a.sh : (u32)MAXB], 1u);
            }
            okm |= nonok ? 0x80000000u : 0u;
            // its number in its bucket's run rides above the count (< 256) from here on
#pragma unroll
            for (int i = 0; i < EPT; ++i) v[i] |= (okm >> i) & 1u ? rk[i] << VBITS : 0u;
        }
        *(u64*)&L.SB[k0] = 0x4040404040404040ull;  // every staging slot: the sink bucket, until P4 fills it
        if (wave == 0 && jrel == 1) {  // window row 0 began before the tile: its entries in front of it
            const u64 x = wave_row_part(rp[2 * jb], apos);
            if (lane == 0) L.head_extra = x;
        }
        if (wave == 1 && tail) {       // a row that began before the tile and runs over its end: its entries behind it
            const u64 x = wave_row_part(anext, rp[2 * (jb + (long long)strad)]);
            if (lane == 0) L.tail_extra = x;
        }
        TS_MARK(1)
        __syncthreads();  // B2
        TS_MARK(6)

        // ---- P3 ------------------------------------------------------------------------------------------
        if (wave == T3W - 1) {  // lane = bucket: the runs of the staging area, chunk ids for what they add
            const u32 m = L.cnt[lane];
            const u32 inc = dme_wave_incl_scan_u32(m);
            if (lane == 63) L.staged = inc;
            L.cnt[lane] = 0;
            const u32 lo = L.len[lane];
            const u32 curold = L.cur[lane];
            const u32 firstnew = (lo + CHUNK - 1) >> 10;            // chunks the stream has opened so far
            // (a stream without a chunk, which only a pool that ran out can cause, is sent to the dump area)
            u32 d0 = curold == 0xFFFFFFFFu ? dump + (inc - m) : (curold << 10) + (lo & (CHUNK - 1));
            u32 d1 = dump + (inc - m);
            const u32 split = (firstnew << 10) - lo;                // 0 when the open chunk is full (or none is open)
            if (m > 0) {
                const u32 lastcn = (lo + m - 1) >> 10;              // chunk number of the run's last record
                const u32 nnew = lastcn + 1 > firstnew ? lastcn + 1 - firstnew : 0u;
                if (nnew) {
                    const long long id0 = cbase + (long long)atomicAdd(&L.wg_next, nnew);
                    u32 lastid = 0xFFFFFFFFu;
                    if (id0 + nnew <= climit) {
                        d1 = ((u32)id0 << 10) - split;
                        lastid = (u32)(id0 + nnew - 1);
                        L.nch[lane] += nnew;
                        for (u32 q = 0; q < nnew; ++q) {
                            a.chunk_bucket[id0 + q] = (unsigned char)lane;
                            a.chunk_fill[id0 + q] = (unsigned short)CHUNK;  // the stream's last one: corrected at the end
                        }
                    } else {
                        atomicAdd(&L.stat[6], 1ULL);  // cannot happen with a well-formed rowptr: the range is sized for S entries
                    }
                    L.cur[lane] = lastid;
                }
                L.len[lane] = lo + m;
            }
            L.prm[lane] = u32x4v{inc - m, split, d0, d1};
            if (lane == 0) L.prm[MAXB] = u32x4v{0u, 0u, dump, dump};  // the sink bucket: slot i goes to dump + i
        }
        if (vrow && (owned || (tid == 0 && jrel == 1))) {  // row totals (DE/dme.py:121 sum(row))
            u64 total;
            if (!has_huge) {
                const int sa = s_rel < 0 ? 0 : (s_rel > n_ent ? n_ent : s_rel);
                const int ta_ = t_rel < 0 ? 0 : (t_rel > n_ent ? n_ent : t_rel);
                total = (u64)(L.PX[ta_] - L.PX[sa]);
                if (tid == 0 && jrel == 1) total += L.head_extra;
                if (tail && strad == (u32)tid + 1) total += L.tail_extra;
            } else {  // a count of 2^19 or more somewhere in the tile: this row's counts one by one, in 64 bits
                total = 0;
                const u32 lo = rp[2 * (jb + tid)], hi = rp[2 * (jb + tid + 1)];
                for (u64 q = lo; q < (u64)hi; ++q) {
                    const int x = a.vals[q];
                    total += (u64)(x > 0 ? x : 0);
                }
            }
            L.RT[tid] = total;
            L.RT32[tid] = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)total;
            if (owned && total > 0) {
                if (total < (u64)a.TB) {
                    atomicAdd(&tb[total], 1u);
                    maxt32 = (u32)total > maxt32 ? (u32)total : maxt32;
                } else {
                    atomicMax(&L.stat[5], total);
                    if ((long long)total < a.tbins) atomicAdd(&a.histV[total], 1ULL);
                    else atomicAdd(&L.stat[2], 1ULL);
                }
            }
        }
        TS_MARK(2)
        __syncthreads();  // B3
        TS_MARK(6)

        // ---- P4 ------------------------------------------------------------------------------------------
        if (tid == 0) {
            L.straddle = 0;
            L.cut = 0xFFFFFFFFu;
            L.huge = 0;
        }
        {
            // every entry's row: the largest F at or before it - a running maximum inside the thread, a DPP
            // max-scan over the wave's threads, and the row the wave starts out in
            u32 f[EPT];
            {
                const u32x4v* fp = (const u32x4v*)&L.F[k0];
                const u32x4v f0 = fp[0], f1 = fp[1];
                u32x4v* fz = (u32x4v*)&L.F[k0];
                fz[0] = u32x4v{0u, 0u, 0u, 0u};  // for the next tile (its rows write behind two more barriers)
                fz[1] = u32x4v{0u, 0u, 0u, 0u};
                f[0] = f0.x;
                f[1] = max(f[0], f0.y);
                f[2] = max(f[1], f0.z);
                f[3] = max(f[2], f0.w);
                f[4] = max(f[3], f1.x);
                f[5] = max(f[4], f1.y);
                f[6] = max(f[5], f1.z);
                f[7] = max(f[6], f1.w);
            }
            const u32 wexcl = max(wave_shr1_u32(wave_incl_scan_max_u32(f[7])), uni(L.WR[wave]));
            // the plain path, eight entries side by side: row total and run offset from LDS, one wait, the
            // records to the staging area; entries it cannot take are only marked
            u32 tot[EPT], ofs[EPT];
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                u32 ri = max(f[i], wexcl);
                ri = ri > 0 ? ri - 1 : 0u;  // entries in front of the first row start: window row 0
                f[i] = ri;
                tot[i] = L.RT32[ri];
                ofs[i] = L.prm[(c[i] >> a.sh) & (MAXB - 1)].x;
            }
            u32 tmx = 0;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const bool ok = (okm >> i) & 1u;
                const bool fits = tot[i] < TLIM;
                const u32 cnt8 = v[i] & (VLIM - 1), rank = v[i] >> VBITS;  // of an entry with `ok`
                const u32 rec = fits ? (c[i] & cmask) | (cnt8 << COLBITS) | (tot[i] << (COLBITS + VBITS)) : 0u;
                if (ok) {
                    L.PX[ofs[i] + rank] = rec;  // a record with count 0 is skipped by the bucket pass
                    L.SB[ofs[i] + rank] = (unsigned char)(c[i] >> a.sh);
                }
                tmx = tot[i] > tmx ? tot[i] : tmx;
            }
            // an entry that is not staged, or a row total a record cannot hold (of any of the eight rows)
            const bool rare = (okm >> 31) != 0 || tmx >= TLIM;
            if (__any(rare)) {
#pragma unroll 1
                for (int i = 0; i < EPT; ++i) {
                    const bool ok = (okm >> i) & 1u;
                    if (v[i] == 0 || (ok && tot[i] < TLIM)) continue;
                    const u32 cnt = ok ? v[i] & (VLIM - 1) : v[i];
                    if ((int)cnt > 0 && c[i] < (u32)a.K) maxv = cnt > maxv ? cnt : maxv;
                    const int r = csr3_rare_entry(a.hist, a.init_acc, a.nbins, a.K, c[i], cnt, L.RT[f[i]], a.overflow);
                    if (r == 1) atomicAdd(&L.stat[3], 1ULL);
                    if (r == 2) atomicAdd(&L.stat[1], 1ULL);
                }
            }
        }
        TS_MARK(3)
        __syncthreads();  // B4
        TS_MARK(6)

        // ---- P5: the runs leave for the buckets' streams -----------------------------------------------------
        {
            // thread t takes the staged records t, t + 512, ...: neighbours in a run are neighbours in the
            // bucket's stream, so a wave's store covers a few contiguous pieces
            if (tid < T3W) L.WR[tid] = 0;  // read in P4, written again in the next tile's P2
            u32 t2 = (u32)tid;  // opaque: keeps the eight positions from being held (and spilled) across the loop
            asm volatile("" : "+v"(t2));
            u32 rec[EPT], bk[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const u32 pos = (u32)(k * T3) + t2;
                rec[k] = L.PX[pos];
                bk[k] = L.SB[pos];
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const u32 pos = (u32)(k * T3) + t2;
                const u32x4v pr = L.prm[bk[k]];
                const u32 i = pos - pr.x;
                a.chunks[i < pr.y ? pr.z + i : pr.w + i] = rec[k];
            }
        }
        TS_MARK(4)
        jown = jownN;
        if (!cont) break;
        apos = anext;
        jb = jbN;
    }

    TS_FLUSH(ta.ts)
    // ---- the streams' last chunks, the statistics of the pass ----------------------------------------------
    __syncthreads();
    if (tid < a.NB) {
        const u32 n = L.len[tid], id = L.cur[tid];
        if (n > 0 && (n & (CHUNK - 1)) && id != 0xFFFFFFFFu) a.chunk_fill[id] = (unsigned short)(n & (CHUNK - 1));
        // the chunk sort's counts: every chunk taken holds records (768 workgroups x 49 adds at their own
        // ends, instead of a counting pass over the whole chunk table behind this kernel: 22 us + a memset)
        if (L.nch[tid]) atomicAdd(&ta.bcount[tid], L.nch[tid]);
    }
    const u32 mv = dme_wave_max_u32(maxv), mt = dme_wave_max_u32(maxt32);
    if (lane == 0) {
        if (mv) atomicMax(&L.stat[4], (u64)mv);
        if (mt) atomicMax(&L.stat[5], (u64)mt);
    }
    if (tid == 0 && jown > jown_first) atomicAdd(&L.stat[0], (u64)(jown - jown_first));  // rows this workgroup owned
    __syncthreads();
    for (int i = tid; i < a.TB; i += T3) {
        const u32 cn = tb[i];
        if (cn) atomicAdd(&a.histV[i], (u64)cn);
    }
    if (tid <= 3 && L.stat[tid]) atomicAdd(&a.status[tid], L.stat[tid]);
    if (tid >= 4 && tid <= 5 && L.stat[tid]) atomicMax(&a.status[tid], L.stat[tid]);
    if (tid == 6 && L.stat[6]) atomicAdd(&a.status[6], L.stat[6]);
}

// Lists the chunks bucket by bucket (a counting sort over three small launches):
// sorted[i] = chunk id | (records in the chunk - 1) << IDBITS for the chunks that hold records,
// bstart[b] = first position of bucket b in `sorted`, bstart[NB] = number of listed chunks.
constexpr int IDBITS = 21;           // chunk ids below 2^21 (the plan refuses larger pools)
constexpr int SBLOCK = 256;
constexpr int SGRID = 1024;         // (128 workgroups measured the same: 20 + 25 us against 22 + 21 us for the two launches)

__global__ __launch_bounds__(SBLOCK) void csr2_sort_count_kernel(long long max_chunks,
                                                                 const unsigned char* chunk_bucket,
                                                                 const unsigned short* chunk_fill,
                                                                 u32* bcount /*[MAXB], zeroed*/,
                                                                 const u32* rowptr_bad) {
    // (the tile kernel has counted its chunks itself; this pass is for the row-wise kernel, which only
    // runs when rowptr is malformed - uniform over the grid)
    if (*rowptr_bad == 0) return;
    __shared__ u32 cnt[MAXB];
    const int tid = threadIdx.x;
    if (tid < MAXB) cnt[tid] = 0;
    __syncthreads();
    const long long nc = max_chunks;  // ids nobody took have fill 0 (the table is zeroed per call)
    for (long long c = (long long)blockIdx.x * SBLOCK + tid; c < nc; c += (long long)SGRID * SBLOCK)
        if (chunk_fill[c] > 0) atomicAdd(&cnt[chunk_bucket[c] & (MAXB - 1)], 1u);
    __syncthreads();
    if (tid < MAXB && cnt[tid]) atomicAdd(&bcount[tid], cnt[tid]);
}

__global__ __launch_bounds__(64) void csr2_sort_scan_kernel(const u32* bcount, int NB, u32* bstart,
                                                            u32* bpos) {
    const int lane = threadIdx.x;
    const u32 x = lane < NB ? bcount[lane] : 0u;
    u32 inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane < NB) {
        bstart[lane] = inc - x;
        bpos[lane] = inc - x;
    }
    if (lane == 63) bstart[NB] = inc;
}

__global__ __launch_bounds__(SBLOCK) void csr2_sort_fill_kernel(long long max_chunks,
                                                                const unsigned char* chunk_bucket,
                                                                const unsigned short* chunk_fill,
                                                                u32* bpos, u32* sorted) {
    __shared__ u32 cnt[MAXB];
    __shared__ u32 first[MAXB];
    const int tid = threadIdx.x;
    const long long nc = max_chunks;
    // a workgroup's chunks: the contiguous slice [lo, hi)
    const long long per = (nc + (long long)gridDim.x - 1) / (long long)gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    long long hi = lo + per;
    if (hi > nc) hi = nc;
    if (tid < MAXB) cnt[tid] = 0;
    __syncthreads();
    for (long long c = lo + tid; c < hi; c += SBLOCK)
        if (chunk_fill[c] > 0) atomicAdd(&cnt[chunk_bucket[c] & (MAXB - 1)], 1u);
    __syncthreads();
    if (tid < MAXB) {
        first[tid] = cnt[tid] ? atomicAdd(&bpos[tid], cnt[tid]) : 0u;
        cnt[tid] = 0;
    }
    __syncthreads();
    for (long long c = lo + tid; c < hi; c += SBLOCK) {
        const u32 f = chunk_fill[c];
        if (f > 0) {
            const int b = chunk_bucket[c] & (MAXB - 1);
            sorted[first[b] + atomicAdd(&cnt[b], 1u)] = (u32)c | ((f - 1) << IDBITS);
        }
    }
}

struct BucketArgs {
    const u32* chunks;
    const u32* sorted;  // chunk id | (fill - 1) << IDBITS
    const u32* bstart;  // [NB + 1]
    long long K, nbins;
    int sh, NB, BL, W, AC;
    u64* hist;
    double* init_acc;
    const u64* status;
    // [NB + W] LDS images: AC x CB fp64 accumulators, BL x CB counters, then the overflow table
    // (OVF keys, OVF counts) for counts beyond the bins
    char* images;
    long long image_bytes;
};

__device__ __forceinline__ u32 piece_chunks(u32 total, int W) { return (total + (u32)W - 1) / (u32)W; }

// Frequent columns.  Word-frequency data puts a few columns in most rows (the three most frequent words
// of BASELINE configs[3] stand in 90-100 % of the documents): half the records of such a column's bucket
// carry the same column, and LDS atomics on one address take the lanes one after the other (2 cycles a
// lane for ds_add_u32, 3 for ds_add_f64: benchmarks/micro/lds_atomics.hip) - the pieces of those buckets
// ran twice as long as the rest (1.75 ms for the shard against 0.85 ms for data without frequent words).
// Every run therefore looks at its first chunk, takes the HOT most frequent columns of it, and keeps them
// out of the shared cells: their mean accumulators are registers of every thread (summed over the
// workgroup at the end of the run), their counts go to a table of their own with HC copies of every bin,
// the lane's copy chosen by its number.  Which columns are picked changes the speed only.
#ifndef DME_CSR2_HOT
#define DME_CSR2_HOT 2
#endif
constexpr int HOT = DME_CSR2_HOT;   // frequent columns per run
constexpr int HC = 8;    // copies of a frequent column's bins
constexpr int HOT_MIN = 16;  // occurrences in the first chunk that make a column frequent
constexpr int HOT_LDS_BYTES = HOT * (int)VLIM * HC * 4 + 64;

__global__ __launch_bounds__(BBLOCK) void csr2_bucket_kernel(BucketArgs a) {
    extern __shared__ u64 lds64[];
    const int CB = 1 << a.sh;
    double* acc = (double*)lds64;        // [AC][CB]
    u32* h = (u32*)(acc + a.AC * CB);    // [BL][CB]
    // Counts beyond the bins belong to a few head-of-vocabulary columns, whose (column, count) cells
    // repeat: sent to global atomics they pile up on a few addresses, which the memory side serialises
    // (3.3 ms for 189 M records, most of it waiting on ~1 % of them).  They are counted in a small
    // open-addressing table of the image instead: key = column-in-bucket << VBITS | count, + 1.
    u32* okey = h + CB * a.BL;           // [OVF]
    u32* ocnt = okey + OVF;              // [OVF]
    // 1 / (1 + n) for row totals below WTAB (DE/fdp.py:88 weightForMean), behind the image: the fp64
    // division was a third of the pass's vector instructions
    double* wtab = (double*)(ocnt + OVF);
    u32* hh = (u32*)(wtab + WTAB);       // [HOT][VLIM][HC] bins of the run's frequent columns
    u32* top = hh + HOT * VLIM * HC;     // [HOT] count << 10 | column of the first chunk's most frequent columns
    for (int n = threadIdx.x; n < WTAB; n += BBLOCK) wtab[n] = __ddiv_rn(1.0, __dadd_rn(1.0, (double)n));
    const u32 acmask = (u32)a.AC - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 total = a.bstart[a.NB];
    const u32 piece = piece_chunks(total, a.W);
    u32 pos = piece * blockIdx.x;
    u32 end = pos + piece;
    if (end > total) end = total;
    if (pos >= end) return;
    int b = 0;
    while (a.bstart[b + 1] <= pos) ++b;  // NB <= 64
    const int cells = CB * a.BL;
    const u32 cmask = (1u << COLBITS) - 1;
    // one (column, count) cell with `n` rows into the image: a bin, or the overflow table
    auto add_cell = [&](u32 cl, u32 v, u32 n, long long kbase) {
        if (v <= (u32)a.BL) {
            atomicAdd(&h[(v - 1) * CB + cl], n);  // bin-major: bank = column
            return;
        }
        const u32 key = ((cl << VBITS) | v) + 1u;
        u32 slot = (key * 2654435761u) >> (32 - 11);  // OVF = 2^11
        bool placed = false;
        for (int probe = 0; probe < 16 && !placed; ++probe) {
            const u32 old = atomicCAS(&okey[slot], 0u, key);
            if (old == 0u || old == key) {
                atomicAdd(&ocnt[slot], n);
                placed = true;
            }
            slot = (slot + 1) & (OVF - 1);
        }
        if (!placed) atomicAdd(&a.hist[(kbase + cl) * a.nbins + v], (u64)n);  // table crowded
    };
    while (pos < end) {
        u32 seg_end = a.bstart[b + 1];
        if (seg_end > end) seg_end = end;
        if (seg_end > pos) {
            for (int i = tid; i < cells + 2 * OVF; i += BBLOCK) h[i] = 0;  // bins and overflow table
            for (int i = tid; i < a.AC * CB; i += BBLOCK) acc[i] = 0.0;
            for (int i = tid; i < HOT * (int)VLIM * HC + HOT; i += BBLOCK) hh[i] = 0;  // and top[]
            __syncthreads();
            const long long kbase = (long long)b << a.sh;
            // ---- the run's frequent columns: the first chunk counted by column in the first bin row ----------
            u32 hotc[HOT];
            {
                const u32 se = a.sorted[pos];
                const u32 rec = (u32)tid <= (se >> IDBITS) ? a.chunks[(long long)(se & ((1u << IDBITS) - 1)) * CHUNK + tid] : 0u;
                if ((rec >> COLBITS) & (VLIM - 1)) atomicAdd(&h[rec & cmask], 1u);
                __syncthreads();
                u32 key = tid < CB ? (h[tid] << COLBITS) | (u32)tid : 0u;
                __syncthreads();
                if (tid < CB) h[tid] = 0;
#pragma unroll
                for (int k = 0; k < HOT; ++k) {  // the largest key, then the largest of the rest, ...
                    const u32 m = dme_wave_max_u32(key);
                    if (lane == 0 && m) atomicMax(&top[k], m);
                    __syncthreads();
                    const u32 t = top[k];
                    hotc[k] = (t >> COLBITS) >= (u32)HOT_MIN ? t & cmask : 0xFFFFFFFFu;
                    if (key == t) key = 0;
                }
            }
            double ah[HOT];  // this thread's share of the frequent columns' mean accumulators
#pragma unroll
            for (int k = 0; k < HOT; ++k) ah[k] = 0.0;
            auto hot_index = [&](u32 cl) -> int {
                int r = -1;
#pragma unroll
                for (int k = 0; k < HOT; ++k) r = cl == hotc[k] ? k : r;
                return r;
            };
            // Counts beyond the bins and row totals beyond the weight table: one record at a time,
            // only in waves that hold such a record (head-of-vocabulary columns, very long rows).
            auto rare_one = [&](u32 rec) {
                const u32 v = (rec >> COLBITS) & (VLIM - 1);
                if (v == 0) return;
                const u32 cl = rec & cmask;
                const u32 tot = rec >> (COLBITS + VBITS);
                const bool hot = hot_index(cl) >= 0;
                if (v > (u32)a.BL && !hot) add_cell(cl, v, 1u, kbase);
                if (tot >= (u32)WTAB) {
                    const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)tot));
                    __hip_atomic_fetch_add(&acc[((u32)tid & acmask) * CB + cl], __dmul_rn((double)v, w),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            // a chunk is CHUNK = BBLOCK records: one per thread.  Two register buffers of eight chunks:
            // the loads of the next eight are in flight while the current eight are counted.  The eight
            // records of a buffer go through the LDS pipe side by side (eight weight reads, one wait,
            // eight counter adds, eight accumulator adds): taken one record at a time, every record paid
            // the latency of its own weight read.
            constexpr int D = 8;
            u32 vzero;  // a zero the compiler cannot see: keeps the chunk list's loads vector loads
            asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
            auto fetch = [&](u32 ci, u32 (&rec)[D]) {
                u32 se[D];
#pragma unroll
                for (int q = 0; q < D; ++q) se[q] = ci + q < seg_end ? a.sorted[ci + q + vzero] : 0u;
#pragma unroll
                for (int q = 0; q < D; ++q)
                    rec[q] = (ci + q < seg_end && (u32)tid <= (se[q] >> IDBITS))
                                 ? __builtin_nontemporal_load(a.chunks + (long long)(se[q] & ((1u << IDBITS) - 1)) * CHUNK + tid)
                                 : 0u;
            };
            double* const acc_mine = acc + ((u32)tid & acmask) * CB;
            u32* const hh_mine = hh + (lane & (HC - 1));
            auto count = [&](const u32 (&rec)[D]) {
                u32 v[D], cl[D];
                double w[D];
                bool rare = false;
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    v[q] = (rec[q] >> COLBITS) & (VLIM - 1);
                    cl[q] = rec[q] & cmask;
                    const u32 tot = rec[q] >> (COLBITS + VBITS);
                    w[q] = wtab[tot < (u32)WTAB ? tot : 0u];
                }
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    const int hi = hot_index(cl[q]);
                    const bool hot = hi >= 0;
                    const bool intab = (rec[q] >> (COLBITS + VBITS)) < (u32)WTAB;
                    rare |= (v[q] > (u32)a.BL && !hot) || !intab;
                    const double x = intab ? __dmul_rn((double)v[q], w[q]) : 0.0;
                    if (hot) {  // count 0 (a slot beyond the chunk's fill) lands in bin 0 of the table, which nobody reads
                        atomicAdd(&hh_mine[((u32)hi * VLIM + v[q]) * HC], 1u);
                    } else {
                        if (v[q] - 1u < (u32)a.BL) atomicAdd(&h[(v[q] - 1) * CB + cl[q]], 1u);  // bin-major: bank = column
                        if (v[q] != 0 && intab)
                            __hip_atomic_fetch_add(&acc_mine[cl[q]], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
#pragma unroll
                    for (int k = 0; k < HOT; ++k) ah[k] = __dadd_rn(ah[k], hi == k ? x : 0.0);
                }
                if (__any(rare)) {
#pragma unroll
                    for (int q = 0; q < D; ++q) rare_one(rec[q]);
                }
            };
            u32 ra[D], rb[D];
            u32 ci = pos;
            fetch(ci, ra);
            while (true) {
                fetch(ci + D, rb);
                count(ra);
                ci += D;
                if (ci >= seg_end) break;
                fetch(ci + D, ra);
                count(rb);
                ci += D;
                if (ci >= seg_end) break;
            }
            // ---- the frequent columns join the image ------------------------------------------------------------
#pragma unroll
            for (int k = 0; k < HOT; ++k) ah[k] = dme_wave_sum_f64(ah[k]);
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < HOT; ++k)
                    if (hotc[k] != 0xFFFFFFFFu && ah[k] != 0.0)
                        __hip_atomic_fetch_add(&acc[hotc[k]], ah[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            for (int t = tid; t < HOT * (int)VLIM; t += BBLOCK) {
                const u32 v = (u32)t & (VLIM - 1);
                u32 col = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < HOT; ++k) col = (t >> VBITS) == k ? hotc[k] : col;
                u32 n = 0;
#pragma unroll
                for (int q = 0; q < HC; ++q) n += hh[t * HC + q];
                if (n && v && col != 0xFFFFFFFFu) add_cell(col, v, n, kbase);
            }
            __syncthreads();
            // The run (b, piece) is the (b + piece)-th of the list: runs are ordered and each one
            // advances b or the piece.  Its LDS image goes out as it is.
            {
                uint4* dst = (uint4*)(a.images + (long long)(b + (int)blockIdx.x) * a.image_bytes);
                const uint4* src = (const uint4*)lds64;
                const int n16 = (int)(a.image_bytes >> 4);
                for (int i = tid; i < n16; i += BBLOCK) dst[i] = src[i];
            }
            __syncthreads();
        }
        pos = seg_end;
        ++b;
    }
}

// hist[k][1..BL] += the counters of every run of bucket b; init_acc[k] += its accumulators.
// grid (NB, ceil(CB / RCT)): a block takes RCT = 32 columns of the bucket and all their bins.  The images
// are bin-major (a bin's columns side by side), hist is column-major (a column's bins side by side): the
// sums of 32 bins x 32 columns cross over through LDS, so that the images are read in 128-byte pieces and
// hist is updated in 256-byte pieces (with a thread per image cell, consecutive threads updated hist
// 8 bytes at a time at a stride of nbins * 8 bytes - 1.25 M scattered read-modify-writes for the shard of
// BASELINE configs[3]: 82 us).
constexpr int RCT = 32;
__global__ __launch_bounds__(BBLOCK) void csr2_reduce_kernel(BucketArgs a) {
    static_assert(BBLOCK == 32 * RCT, "a block is 32 bins x RCT columns");
    __shared__ u64 tile[32][RCT + 1];
    const int b = blockIdx.x, tid = threadIdx.x;
    const u32 s0 = a.bstart[b], s1 = a.bstart[b + 1];
    if (s1 == s0) return;
    const int CB = 1 << a.sh;
    const u32 piece = piece_chunks(a.bstart[a.NB], a.W);
    const int w_lo = (int)(s0 / piece), w_hi = (int)((s1 - 1) / piece);
    const long long kbase = (long long)b << a.sh;
    const int cells = CB * a.BL;
    {
        const int c0 = (int)blockIdx.y * RCT;            // first column (inside the bucket) of this block
        const int ct = CB - c0 < RCT ? CB - c0 : RCT;    // (< RCT only for buckets of fewer than RCT columns)
        const int lc = tid & (RCT - 1), lb = tid / RCT;  // reading: column lc of bin lb of a group of 32 bins
        const int wc = tid >> 5, wb = tid & 31;          // writing: bin wb of column wc
        for (int bin0 = 0; bin0 < a.BL; bin0 += 32) {
            const int bin = bin0 + lb;                   // 0-based: counts of bin + 1
            u64 sum = 0;
            if (bin < a.BL && lc < ct) {
                const char* img = a.images + 8LL * a.AC * CB + 4LL * ((long long)bin * CB + c0 + lc);
                int w = w_lo;
                for (; w + 7 <= w_hi; w += 8) {
                    u32 x[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[q] = *(const u32*)(img + (long long)(b + w + q) * a.image_bytes);
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += x[q];
                }
                for (; w <= w_hi; ++w) sum += *(const u32*)(img + (long long)(b + w) * a.image_bytes);
            }
            tile[lb][lc] = sum;
            __syncthreads();
            const u64 v = tile[wb][wc];
            if (v && wc < ct && bin0 + wb < a.BL && kbase + c0 + wc < a.K)
                a.hist[(kbase + c0 + wc) * a.nbins + bin0 + wb + 1] += v;
            __syncthreads();
        }
    }
    // the overflow tables of the bucket's runs: block y takes the runs w = w_lo + y, + gridDim.y, ...
    for (int w = w_lo + (int)blockIdx.y; w <= w_hi; w += (int)gridDim.y) {
        const u32* okey = (const u32*)(a.images + (long long)(b + w) * a.image_bytes + 8LL * a.AC * CB + 4LL * cells);
        const u32* ocnt = okey + OVF;
        for (int sl = tid; sl < OVF; sl += BBLOCK) {
            const u32 key = okey[sl];
            if (key) {
                const u32 cl = (key - 1) >> VBITS, v = (key - 1) & (VLIM - 1);
                if (kbase + cl < a.K && (long long)v < a.nbins)
                    atomicAdd(&a.hist[(kbase + cl) * a.nbins + v], (u64)ocnt[sl]);
            }
        }
    }
    for (int c = tid; c < CB; c += BBLOCK) {
        double sum = 0.0;
        for (int w = w_lo + (int)blockIdx.y; w <= w_hi; w += (int)gridDim.y) {
            const double* img = (const double*)(a.images + (long long)(b + w) * a.image_bytes);
            for (int q = 0; q < a.AC; ++q) sum = __dadd_rn(sum, img[q * CB + c]);
        }
        if (sum != 0.0 && kbase + c < a.K) atomicAdd(&a.init_acc[kbase + c], sum);
    }
}

}  // namespace dme_csr2

// Entry points for dme_csr.hip ---------------------------------------------------------------------
long long dme_csr2_workspace_bytes(long long N, long long nnz, long long K) {
    dme_csr2::Plan p;
    if (!dme_csr2::plan(N, nnz, K, 1LL << 30, &p)) return -1;  // the largest LDS image (nbins unknown here)
    return p.bytes;
}

// Returns DME_OK, an error, or -1 when the shape is not this design's (more than 65 536 columns).
int dme_csr2_run(const long long* rowptr, const int* colidx, const int* vals, long long N, long long nnz,
                 long long K, u64* hist, long long nbins, u64* histV, long long tbins, double* init_acc,
                 u64* status, u64* overflow, void* workspace, long long workspace_bytes, hipStream_t s) {
    using namespace dme_csr2;
    Plan p;
    if (!plan(N, nnz, K, nbins, &p)) return -1;
    DME_REQUIRE(workspace_bytes >= p.bytes && ((uintptr_t)workspace & 255) == 0,
                "CSR workspace: %lld bytes (256-byte aligned) needed, got %lld", p.bytes, workspace_bytes);
    char* ws = (char*)workspace;
    PartArgs a{};
    a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
    a.N = N; a.nnz = nnz; a.K = K; a.nbins = nbins; a.tbins = tbins;
    a.rows_per_wg = p.rows_per_wg; a.max_chunks = p.max_chunks;
    a.sh = p.sh; a.NB = p.NB;
    a.TB = (int)(tbins < TBL ? tbins : TBL);
    a.chunks_per_wg_extra = (long long)PWAVES * p.NB + 2;
    a.chunk_bucket = (unsigned char*)(ws + p.off_cbucket);
    a.chunk_fill = (unsigned short*)(ws + p.off_cfill);
    a.chunks = (u32*)(ws + p.off_chunks);
    a.hist = hist; a.histV = histV; a.init_acc = init_acc; a.status = status; a.overflow = overflow;
    DME_HIP_CHECK(hipMemsetAsync(a.chunk_fill, 0, 2 * p.max_chunks, s));
    // flag and rstart lie side by side: one memset
    DME_HIP_CHECK(hipMemsetAsync(ws + p.off_flag, 0, (size_t)(p.off_rstart - p.off_flag) + 8 * ((size_t)p.G3 + 1), s));
    TileArgs ta{};
    ta.flag = (u32*)(ws + p.off_flag);
    ta.ts = (u64*)(ws + p.off_head);
#ifdef DME_CSR3_TS
    DME_HIP_CHECK(hipMemsetAsync(ws + p.off_head, 0, 256, s));
#endif
    ta.rstart = (long long*)(ws + p.off_rstart);
    ta.bcount = (u32*)(ws + p.off_bcount);
    ta.G = p.G3;
    ta.chunks_per_wg = p.chunks_per_wg3;
    a.rowptr_bad = ta.flag;
    ta.a = a;
    const size_t lds_tile = sizeof(TileLds) + 4 * (size_t)a.TB;
    const size_t lds_part = sizeof(WaveStage) * PWAVES + 4 * (size_t)a.TB;
    static std::atomic<bool> attr_set[64];  // zero-initialised; racing first calls both set the attribute
    int dev = 0;
    DME_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)csr2_part_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(sizeof(WaveStage) * PWAVES + 4 * TBL)));
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)csr2_bucket_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          IMAGE_LDS_BYTES + 8 * WTAB + HOT_LDS_BYTES));
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)csr3_part_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(sizeof(TileLds) + 4 * TBL)));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    {
        long long cg = (N + 255) / 256;
        if (cg > 2048) cg = 2048;
        hipLaunchKernelGGL(csr3_check_kernel, dim3((unsigned)cg), dim3(256), 0, s, rowptr, N, nnz, p.G3, ta.flag, ta.rstart);
    }
    hipLaunchKernelGGL(csr3_part_kernel, dim3(p.G3), dim3(T3), lds_tile, s, ta);
    hipLaunchKernelGGL(csr2_part_kernel, dim3(p.G), dim3(PBLOCK), lds_part, s, a);  // leaves at once unless rowptr is malformed
    u32* sorted = (u32*)(ws + p.off_sorted);
    u32* bstart = (u32*)(ws + p.off_bstart);
    u32* bcount = (u32*)(ws + p.off_bcount);
    u32* bpos = (u32*)(ws + p.off_bpos);
    hipLaunchKernelGGL(csr2_sort_count_kernel, dim3(SGRID), dim3(SBLOCK), 0, s, p.max_chunks, (const unsigned char*)a.chunk_bucket,
                       (const unsigned short*)a.chunk_fill, bcount, (const u32*)ta.flag);
    hipLaunchKernelGGL(csr2_sort_scan_kernel, dim3(1), dim3(64), 0, s, (const u32*)bcount, p.NB, bstart, bpos);
    hipLaunchKernelGGL(csr2_sort_fill_kernel, dim3(SGRID), dim3(SBLOCK), 0, s, p.max_chunks, (const unsigned char*)a.chunk_bucket,
                       (const unsigned short*)a.chunk_fill, bpos, sorted);
    BucketArgs bk{};
    bk.chunks = a.chunks; bk.sorted = sorted; bk.bstart = bstart;
    bk.K = K; bk.nbins = nbins; bk.sh = p.sh; bk.NB = p.NB; bk.BL = p.BL; bk.W = p.W; bk.AC = p.AC;
    bk.hist = hist; bk.init_acc = init_acc; bk.status = status;
    bk.images = ws + p.off_images;
    bk.image_bytes = p.image_bytes;
    hipLaunchKernelGGL(csr2_bucket_kernel, dim3(p.W), dim3(BBLOCK), (size_t)p.image_bytes + 8 * WTAB + HOT_LDS_BYTES, s, bk);
    hipLaunchKernelGGL(csr2_reduce_kernel, dim3(p.NB, (p.CB + RCT - 1) / RCT), dim3(BBLOCK), 0, s, bk);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
