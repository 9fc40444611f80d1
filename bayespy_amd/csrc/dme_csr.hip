// dme_csr.hip — statistics pass over CSR rows (bag-of-words shapes: K far beyond what one
// workgroup's LDS can privatise).
//
// Reference semantics: CompressedRowData.appendRow over every row (DE/dme.py:113-123) and the CLI's
// mean accumulator (DE/fdp.py:87-89), as for the dense kernels: hist[k][v] += 1 for every stored
// entry (k, v > 0), histV[n_i] += 1, init_acc[k] += v / (1 + n_i).
//
// Scattered 64-bit global atomics (two per stored entry) run at ~7 G atomics/s on MI355X — 0.4 %
// of the HBM roofline on Zipf-distributed words.  The partitioned path removes them:
//
//   csr_count_kernel    each workgroup owns a contiguous slab of rows; counts its entries per
//                       column bucket (bucket = CB consecutive columns) in LDS        [reads colidx]
//   csr_offsets_kernel  + csr_bucket_scan_kernel: bucket-major exclusive offsets of every
//                       (bucket, slab) run
//   csr_scatter_kernel  batches of whole rows, entry-parallel: row totals and row ids from LDS
//                       prefix sums -> histV (LDS bins);
//                       every entry becomes 8 bytes {column-in-bucket | count << cb, row total},
//                       ordered by bucket in LDS, written in contiguous pieces to its bucket's
//                       run                                                 [reads all, writes 8 B]
//   csr_bucket_kernel   the bucket-major entry stream is cut into equal pieces, one per
//                       workgroup; a piece holds one bucket (or a few): counts 1..BL of its CB
//                       columns are privatised in LDS (32-bit), the mean accumulator too (LDS fp64
//                       atomics); the LDS image of every (bucket, piece) run is dumped as it is
//   csr_reduce_kernel   adds the images of each bucket into hist / init_acc (plain +=: a bucket's
//                       columns belong to one workgroup row of the grid)
//
// Integer results are exact and order-independent; init_acc is summed in arrival order (fp64), so
// its last bits vary from run to run, as with the atomics path.  Without a workspace, or for inputs
// too small to repay six launches, the atomics kernel runs.
#include "dme_common.h"

// dme_csr2.hip: the second design, for K <= 65 536 columns
long long dme_csr2_workspace_bytes(long long N, long long nnz, long long K);
int dme_csr2_run(const long long* rowptr, const int* colidx, const int* vals, long long N, long long nnz,
                 long long K, u64* hist, long long nbins, u64* histV, long long tbins, double* init_acc,
                 u64* status, u64* overflow, void* workspace, long long workspace_bytes, hipStream_t s);

namespace {

constexpr int CSR_BLOCK = 1024;
constexpr int CSR_WAVES = CSR_BLOCK / 64;
constexpr int CSR_MAX_BUCKETS = 2048;
constexpr int CSR_HIST_LDS_BYTES = 60 * 1024;
constexpr int CSR_TBL_MAX = 2048;  // row totals below this are counted in LDS
// The scatter kernel runs as many small workgroups (4 waves, ~35 KB of LDS, four per CU): its
// batches are chains of short dependent LDS phases, and independent workgroups overlap them where
// one 1024-thread workgroup per CU would sit in barriers.
constexpr int CSR_SBLOCK = 256;
constexpr int CSR_SWAVES = CSR_SBLOCK / 64;
constexpr int CSR_BATCH = 2048;    // entries per batch
constexpr int CSR_EPT = CSR_BATCH / CSR_SBLOCK;
constexpr int CSR_RB = CSR_SBLOCK;  // rows per batch at most (one prefetched row pointer per thread)
constexpr long long CSR_SMALL_NNZ = 1 << 15;

// ---------------------------------------------------------------------------------------------
// atomics path (small inputs, no workspace)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_csr_kernel(const long long* rowptr, const int* colidx,
                                                        const int* vals, long long N, long long nnz,
                                                        long long K, u64* hist, long long nbins,
                                                        u64* histV, long long tbins, double* init_acc,
                                                        u64* status, u64* overflow) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    __shared__ u64 sh_stat[8];  // [1] cells >= nbins [2] rows >= tbins [3] bad entries [4],[5] maxima
    if (threadIdx.x < 8) sh_stat[threadIdx.x] = 0;
    __syncthreads();
    u64 nrows = 0;
    for (long long i = wave0; i < N; i += nwaves) {
        const long long p0 = rowptr[i], p1 = rowptr[i + 1];
        if (p0 < 0 || p1 < p0 || p1 > nnz) {  // malformed rowptr: the row is not touched (wave-uniform)
            if (lane == 0) atomicAdd(&status[6], 1ULL);
            continue;
        }
        u64 s = 0;
        for (long long p = p0 + lane; p < p1; p += 64) {
            const int v = vals[p];
            s += (u64)(v > 0 ? v : 0);
        }
        const u64 total = dme_wave_sum_u64(s);
        nrows++;
        if (lane == 0 && total > 0) {
            atomicMax(&sh_stat[5], total);
            if ((long long)total < tbins) atomicAdd(&histV[total], 1ULL);
            else atomicAdd(&sh_stat[2], 1ULL);
        }
        const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
        for (long long p = p0 + lane; p < p1; p += 64) {
            const int v = vals[p];
            const long long k = colidx[p];
            if (v == 0) continue;
            if (v < 0 || k < 0 || k >= K) { atomicAdd(&sh_stat[3], 1ULL); continue; }
            atomicMax(&sh_stat[4], (u64)v);
            if ((long long)v < nbins) atomicAdd(&hist[k * nbins + v], 1ULL);
            else if (!dme_overflow_put(overflow, (u64)k, (u64)v, 1ULL)) atomicAdd(&sh_stat[1], 1ULL);
            atomicAdd(&init_acc[k], __dmul_rn((double)v, w));
        }
    }
    if (lane == 0 && nrows) atomicAdd(&status[0], nrows);
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= 1 && t <= 3 && sh_stat[t]) atomicAdd(&status[t], sh_stat[t]);
    if (t >= 4 && t <= 5 && sh_stat[t]) atomicMax(&status[t], sh_stat[t]);
}

// ---------------------------------------------------------------------------------------------
// partitioned path
// ---------------------------------------------------------------------------------------------
// Adds 1 to cnt[b] for every lane with ok set.  Word-frequency data puts most of a wave's lanes in
// one or two buckets (the head of the vocabulary), and same-address LDS atomics serialise: the
// bucket of the first pending lane is counted with one atomic for all its lanes, twice, before the
// remaining lanes go one by one.
__device__ __forceinline__ void csr_count_lanes(u32* cnt, int b, bool ok) {
    u64 pending = __ballot(ok);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (!pending) return;
        const int lead = __builtin_ctzll(pending);
        const int bl = __shfl(b, lead);
        const u64 same = __ballot(ok && b == bl) & pending;
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&cnt[bl], (u32)__builtin_popcountll(same));
        pending &= ~same;
        if (b == bl) ok = false;
    }
    if (ok) atomicAdd(&cnt[b], 1u);
}

// The same with the rank of every lane returned (its position among the entries of its bucket).
__device__ __forceinline__ u32 csr_rank_lanes(u32* cnt, int b, bool ok) {
    u64 pending = __ballot(ok);
    const u64 below = (1ULL << (threadIdx.x & 63)) - 1;
    u32 rank = 0;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (!pending) return rank;
        const int lead = __builtin_ctzll(pending);
        const int bl = __shfl(b, lead);
        const u64 same = __ballot(ok && b == bl) & pending;
        u32 first = 0;
        if ((int)(threadIdx.x & 63) == lead) first = atomicAdd(&cnt[bl], (u32)__builtin_popcountll(same));
        first = __shfl(first, lead);
        if (ok && b == bl) {
            rank = first + (u32)__builtin_popcountll(same & below);
            ok = false;
        }
        pending &= ~same;
    }
    if (ok) rank = atomicAdd(&cnt[b], 1u);
    return rank;
}

struct CsrPlan {
    int sh;       // log2(CB)
    int CB;       // columns per bucket
    int NB;       // buckets
    int BL;       // counts 1..BL live in LDS in the bucket pass
    int G;        // row slabs (= workgroups of the count and scatter kernels)
    int W;        // pieces (= workgroups of the bucket pass)
    int copies;   // LDS counter copies in the count kernel
    int AC;       // copies of the fp64 accumulators in the bucket pass (indexed by row total & (AC-1))
    long long rows_per_slab;
    long long vlim;  // counts below this are packed; others go straight to global atomics
    // workspace offsets (bytes)
    long long off_counts, off_rel, off_tot, off_start, off_images, off_entries, bytes;
};

long long align256(long long x) { return (x + 255) & ~255LL; }

int plan_csr(long long N, long long nnz, long long K, long long nbins, CsrPlan* p) {
    int sh = 8;
    while (((K + (1LL << sh) - 1) >> sh) > CSR_MAX_BUCKETS) ++sh;
    if (sh > 13)
        return dme_set_error(DME_ERR_UNSUPPORTED, "CSR statistics: K=%lld is beyond %d buckets of 8192 "
                             "columns", K, CSR_MAX_BUCKETS);
    p->sh = sh;
    p->CB = 1 << sh;
    p->NB = (int)((K + p->CB - 1) >> sh);
    long long bl = CSR_HIST_LDS_BYTES / (4LL * p->CB);
    if (bl > nbins - 1) bl = nbins - 1;
    if (bl < 1) bl = 1;
    p->BL = (int)bl;
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    long long G = 4LL * cus;  // all resident at once: four scatter workgroups per CU
    if (G > (N + 63) / 64) G = (N + 63) / 64;
    if (G < 1) G = 1;
    p->G = (int)G;
    p->rows_per_slab = (N + G - 1) / G;
    p->W = 4 * cus;
    int copies = 16;
    while (copies > 1 && (long long)copies * p->NB > 8192) copies >>= 1;
    p->copies = copies;
    int ac = 8;
    while (ac > 1 && 8LL * ac * p->CB > 16 * 1024) ac >>= 1;
    p->AC = ac;
    const long long vl = 1LL << (32 - sh);
    p->vlim = nbins < vl ? nbins : vl;
    long long o = 0;
    p->off_counts = o; o = align256(o + 4LL * p->G * p->NB);
    p->off_rel = o;    o = align256(o + 8LL * p->G * p->NB);
    p->off_tot = o;    o = align256(o + 8LL * p->NB);
    p->off_start = o;  o = align256(o + 8LL * (p->NB + 1));
    // image size bound: 8*AC*CB + 4*CB*BL <= 8*AC*CB + CSR_HIST_LDS_BYTES whatever nbins is
    p->off_images = o;
    o = align256(o + (long long)(p->NB + p->W) * (8LL * p->AC * p->CB + CSR_HIST_LDS_BYTES));
    p->off_entries = o; o = align256(o + 8LL * (nnz > 0 ? nnz : 1));
    p->bytes = o;
    return DME_OK;
}

// counts[g][b] = entries of slab g whose column lies in bucket b.
// Before it reads colidx a workgroup checks its slab's row pointers: rowptr[r0] >= 0, non-decreasing
// up to rowptr[r1] <= nnz.  A slab that fails raises status[6] and counts nothing; every later
// kernel of the pass returns at once when status[6] is set, so a malformed rowptr (more entries
// than the caller's nnz, or rows that run backwards) never turns into an out-of-bounds access.
__global__ __launch_bounds__(CSR_BLOCK) void csr_count_kernel(const long long* rowptr,
                                                              const int* colidx, long long N,
                                                              long long nnz, long long K,
                                                              long long rows_per_slab, int sh,
                                                              int NB, int copies, u32* counts,
                                                              u64* status) {
    extern __shared__ u32 cnt[];  // [copies][NB]
    const int tid = threadIdx.x;
    for (int i = tid; i < copies * NB; i += CSR_BLOCK) cnt[i] = 0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > N) r1 = N;
    bool bad = false;
    for (long long r = r0 + tid; r < r1; r += CSR_BLOCK) {
        const long long a0 = rowptr[r], a1 = rowptr[r + 1];
        if (a0 < 0 || a1 < a0 || a1 > nnz) bad = true;
    }
    if (__syncthreads_or(bad ? 1 : 0)) {  // block-uniform
        if (tid == 0) atomicAdd(&status[6], 1ULL);
        for (int b = tid; b < NB; b += CSR_BLOCK) counts[(long long)blockIdx.x * NB + b] = 0;
        return;
    }
    if (r0 < r1) {
        const long long e0 = rowptr[r0], e1 = rowptr[r1];
        u32* mine = cnt + ((tid >> 6) & (copies - 1)) * NB;
        long long e = e0 + tid;
        for (; e + 3LL * CSR_BLOCK < e1; e += 4LL * CSR_BLOCK) {
            const int c0 = __builtin_nontemporal_load(colidx + e);
            const int c1 = __builtin_nontemporal_load(colidx + e + CSR_BLOCK);
            const int c2 = __builtin_nontemporal_load(colidx + e + 2 * CSR_BLOCK);
            const int c3 = __builtin_nontemporal_load(colidx + e + 3 * CSR_BLOCK);
            csr_count_lanes(mine, c0 >> sh, (u64)(long long)c0 < (u64)K);
            csr_count_lanes(mine, c1 >> sh, (u64)(long long)c1 < (u64)K);
            csr_count_lanes(mine, c2 >> sh, (u64)(long long)c2 < (u64)K);
            csr_count_lanes(mine, c3 >> sh, (u64)(long long)c3 < (u64)K);
        }
        // the tail: whole waves stay in the loop so that the ballots see every lane
        for (long long eb = e - tid; eb < e1; eb += CSR_BLOCK) {
            const long long q = eb + tid;
            const int c = q < e1 ? colidx[q] : -1;
            csr_count_lanes(mine, c >> sh, q < e1 && (u64)(long long)c < (u64)K);
        }
    }
    __syncthreads();
    for (int b = tid; b < NB; b += CSR_BLOCK) {
        u32 s = 0;
        for (int c = 0; c < copies; ++c) s += cnt[c * NB + b];
        counts[(long long)blockIdx.x * NB + b] = s;
    }
}

// rel[g][b] = sum_{g' < g} counts[g'][b];  tot[b] = sum_g counts[g][b].
// One workgroup per 64 buckets: lane = bucket, wave = a contiguous chunk of slabs.
__global__ __launch_bounds__(CSR_BLOCK) void csr_offsets_kernel(const u32* counts, int G, int NB,
                                                                u64* rel, u64* tot) {
    __shared__ u64 part[CSR_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const int per = (G + CSR_WAVES - 1) / CSR_WAVES;
    const int g0 = wave * per;
    const int g1 = g0 + per < G ? g0 + per : G;
    u64 s = 0;
    if (b < NB)
        for (int g = g0; g < g1; ++g) s += counts[(long long)g * NB + b];
    part[wave][lane] = s;
    __syncthreads();
    u64 run = 0;
    for (int w = 0; w < wave; ++w) run += part[w][lane];
    if (b < NB) {
        for (int g = g0; g < g1; ++g) {
            rel[(long long)g * NB + b] = run;
            run += counts[(long long)g * NB + b];
        }
        if (wave == CSR_WAVES - 1) {
            // the last wave's running sum covers every slab only if it owns the tail; recompute
            u64 t = 0;
            for (int w = 0; w < CSR_WAVES; ++w) t += part[w][lane];
            tot[b] = t;
        }
    }
}

// start[b] = sum_{b' < b} tot[b'], start[NB] = all entries.  NB <= 2048: one workgroup.
__global__ __launch_bounds__(CSR_BLOCK) void csr_bucket_scan_kernel(const u64* tot, int NB,
                                                                    u64* start) {
    __shared__ u64 wsum[CSR_WAVES];
    const int tid = threadIdx.x;
    // two consecutive buckets per thread, wave scan, then the wave sums
    const int i0 = 2 * tid, i1 = 2 * tid + 1;
    const u64 x0 = i0 < NB ? tot[i0] : 0, x1 = i1 < NB ? tot[i1] : 0;
    u64 s = x0 + x1;
    u64 inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = __shfl_up(inc, d);
        if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    u64 base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    const u64 excl = base + inc - s;
    if (i0 < NB) start[i0] = excl;
    if (i1 < NB) start[i1] = excl + x0;
    if (tid == CSR_BLOCK - 1) start[NB] = base + inc;
}

struct CsrArgs {
    const long long* rowptr;
    const int* colidx;
    const int* vals;
    long long N, K, nbins, tbins, rows_per_slab, vlim;
    u64 capacity;  // records the workspace holds (the caller's nnz)
    int sh, NB, TBL;
    const u64* rel;
    const u64* start;
    uint2* entries;
    u64* hist;
    u64* histV;
    double* init_acc;
    u64* status;
    u64* overflow;  // store for counts >= nbins (dme_common.h), or NULL
};

// LDS carve-up of csr_scatter_kernel (bytes), shared with the host for the launch size.
struct ScatterLds {
    int off_base, off_rowsum, off_stage, off_rp, off_bcnt, off_boff, off_tb, off_bkt, off_misc, bytes;
};
__host__ __device__ inline ScatterLds scatter_lds(int NB, int TBL) {
    ScatterLds L;
    int o = 0;
    auto take = [&](int bytes) {
        const int at = o;
        o += (bytes + 15) & ~15;  // every region 16-byte aligned (128-bit LDS accesses)
        return at;
    };
    L.off_base = take(8 * NB);
    L.off_rowsum = take(8 * CSR_RB);
    L.off_stage = take(8 * CSR_BATCH);
    L.off_rp = take(4 * (CSR_RB + 2));
    L.off_bcnt = take(4 * NB);
    L.off_boff = take(4 * NB);
    L.off_tb = take(4 * TBL);
    L.off_bkt = take(2 * CSR_BATCH);
    L.off_misc = take(256);
    L.bytes = o;
    return L;
}

// Exclusive scan of n counters (in -> out) by a CSR_SBLOCK-thread workgroup; every thread gets the
// total.  Thread t owns the consecutive elements [t*per, (t+1)*per).
__device__ u32 block_excl_scan(const u32* in, u32* out, int n, u32* wsum /*[CSR_SWAVES]*/) {
    const int tid = threadIdx.x;
    const int per = (n + CSR_SBLOCK - 1) / CSR_SBLOCK;
    const int lo = tid * per;
    int hi = lo + per;
    if (hi > n) hi = n;
    u32 s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    u32 inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_up(inc, d);
        if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    u32 before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < CSR_SWAVES; ++w) {
        const u32 t = wsum[w];
        if (w < (tid >> 6)) before += t;
        total += t;
    }
    u32 run = before + inc - s;
    for (int i = lo; i < hi; ++i) {
        const u32 x = in[i];
        out[i] = run;
        run += x;
    }
    __syncthreads();
    return total;
}

// Radix partition of one row slab by column bucket.  The slab is cut into batches of whole rows
// holding at most CSR_BATCH entries; a batch is loaded entry-parallel (coalesced, CSR_EPT loads in
// flight per thread, issued one batch ahead), every entry finds its row by binary search over the
// batch's row pointers (LDS), the row totals are integer LDS atomics, then the entries are ordered
// by bucket inside LDS and written to their buckets' runs in contiguous pieces (8-byte stores
// scattered one by one are bound by the L2 request rate).
__global__ __launch_bounds__(CSR_SBLOCK, 4) void csr_scatter_kernel(CsrArgs a) {
    extern __shared__ u64 lds64[];
    char* lds = (char*)lds64;
    const ScatterLds L = scatter_lds(a.NB, a.TBL);
    u64* base = (u64*)(lds + L.off_base);      // next global slot of this slab in each bucket
    u64* rowsum = (u64*)(lds + L.off_rowsum);  // row totals of the batch
    u64* stage = (u64*)(lds + L.off_stage);    // the batch's records, ordered by bucket
    u32* rp = (u32*)(lds + L.off_rp);          // row starts relative to the batch start
    u32* bcnt = (u32*)(lds + L.off_bcnt);
    u32* boff = (u32*)(lds + L.off_boff);
    u32* tb = (u32*)(lds + L.off_tb);          // row-total bins below TBL
    unsigned short* bkt = (unsigned short*)(lds + L.off_bkt);
    u64* sh_stat = (u64*)(lds + L.off_misc);   // [8]
    u32* wsum = (u32*)(sh_stat + 8);           // [CSR_SWAVES]
    int* sh_nr = (int*)(wsum + 16);

    const int tid = threadIdx.x;
    const long long g = blockIdx.x;
    for (int b = tid; b < a.NB; b += CSR_SBLOCK) base[b] = a.start[b] + a.rel[g * a.NB + b];
    for (int i = tid; i < a.TBL; i += CSR_SBLOCK) tb[i] = 0;
    if (tid < 8) sh_stat[tid] = 0;
    const long long r0 = g * a.rows_per_slab;
    long long r1 = r0 + a.rows_per_slab;
    if (r1 > a.N) r1 = a.N;
    if (r0 >= r1) return;  // uniform: nothing was written yet
    // a slab whose row pointers failed the count kernel's check (or an earlier call on these status
    // words did): nothing is read or written
    if (a.status[6] != 0 || a.start[a.NB] > a.capacity) {
        if (g == 0 && tid == 0 && a.status[6] == 0) atomicAdd(&a.status[6], 1ULL);
        return;
    }
    const u32 cmask = (1u << a.sh) - 1;
    u32 maxv = 0;
    u64 maxt = 0, bad = 0, big = 0;
    // entries of the slab, addressed by 32-bit offsets from a uniform base (the host keeps nnz
    // below 2^32 on this path)
    const long long e_slab = a.rowptr[r0];
    const u32 slab_len = (u32)(a.rowptr[r1] - e_slab);
    const int* cslab = a.colidx + e_slab;
    const int* vslab = a.vals + e_slab;
    __syncthreads();

    int c[CSR_EPT], v[CSR_EPT];   // the loaded entries; then {packed record, bucket}
    u32 n32[CSR_EPT];

    auto fetch_ptr = [&](long long r, int j) -> long long {
        const long long q = r + j;
        return q <= r1 ? a.rowptr[q] : 0x7FFFFFFFFFFFFFFFLL;
    };
    auto fetch_entries = [&](u32 off) {
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            const u32 q = off + tid + j * CSR_SBLOCK;
            const bool in = q < slab_len;
            c[j] = in ? __builtin_nontemporal_load(cslab + q) : 0;
            v[j] = in ? __builtin_nontemporal_load(vslab + q) : 0;
        }
    };
    // c/v/n32 -> c = packed record, v = bucket (or -1: dropped)
    auto classify = [&](u32 nb) {
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            const u32 i = tid + j * CSR_SBLOCK;
            const long long k = c[j];
            const int x = v[j];
            c[j] = 0;
            v[j] = -1;
            if (i >= nb) continue;
            if ((u64)k >= (u64)a.K) {  // not counted by csr_count_kernel either
                if (x != 0) bad++;
                continue;
            }
            if (x < 0) {
                bad++;
            } else if (x > 0) {
                maxv = (u32)x > maxv ? (u32)x : maxv;
                if ((long long)x < a.vlim) {
                    c[j] = (int)(((u32)k & cmask) | ((u32)x << a.sh));  // count 0: skipped downstream
                } else {
                    if ((long long)x < a.nbins) atomicAdd(&a.hist[k * a.nbins + x], 1ULL);
                    else if (!dme_overflow_put(a.overflow, (u64)k, (u64)x, 1ULL)) big++;
                    const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)n32[j]));
                    atomicAdd(&a.init_acc[k], __dmul_rn((double)x, w));
                }
            }
            v[j] = (int)(k >> a.sh);
        }
    };
    // Orders the classified records pk/bk/nn by bucket in LDS and writes them out.  bcnt is zero on
    // entry and on exit.
    auto emit = [&](const int* pk, const int* bk, const u32* nn) {
        u32 rank[CSR_EPT];
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) rank[j] = csr_rank_lanes(bcnt, bk[j], bk[j] >= 0);
        __syncthreads();
        const u32 nvalid = block_excl_scan(bcnt, boff, a.NB, wsum);
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            if (bk[j] < 0) continue;
            const u32 pos = boff[bk[j]] + rank[j];
            stage[pos] = (u64)(u32)pk[j] | ((u64)nn[j] << 32);
            bkt[pos] = (unsigned short)bk[j];
        }
        __syncthreads();
        u64* out = (u64*)a.entries;
        for (u32 i = tid; i < nvalid; i += CSR_SBLOCK) {
            const int b = bkt[i];
            out[base[b] + (i - boff[b])] = stage[i];
        }
        __syncthreads();
        for (int b = tid; b < a.NB; b += CSR_SBLOCK) {
            base[b] += bcnt[b];
            bcnt[b] = 0;
        }
        __syncthreads();
    };
    // One row's total -> histV (LDS bins below TBL), maxima, the rows-beyond-tbins counter.
    auto count_total = [&](u64 t) {
        if (t == 0) return;
        maxt = t > maxt ? t : maxt;
        if (t < (u64)a.TBL) atomicAdd(&tb[t], 1u);
        else if ((long long)t < a.tbins) atomicAdd(&a.histV[t], 1ULL);
        else atomicAdd(&sh_stat[2], 1ULL);
    };

    for (int b = tid; b < a.NB; b += CSR_SBLOCK) bcnt[b] = 0;
    long long r = r0;
    u32 off = 0;  // the batch starts at entry e_slab + off
    long long pre0 = fetch_ptr(r, tid), pre1 = tid == 0 ? fetch_ptr(r, CSR_SBLOCK) : 0;
    fetch_entries(0);
    while (r < r1) {
        const int nrw = (int)(r1 - r < CSR_RB ? r1 - r : CSR_RB);
        {
            const long long e0 = e_slab + off;
            const long long d = pre0 - e0;
            rp[tid] = d > 0xFFFFFFFFLL ? 0xFFFFFFFFu : (u32)d;
            if (tid == 0) {
                const long long d1 = pre1 - e0;
                rp[CSR_SBLOCK] = d1 > 0xFFFFFFFFLL ? 0xFFFFFFFFu : (u32)d1;
            }
        }
        __syncthreads();
        if (tid <= nrw && rp[tid] <= (u32)CSR_BATCH && (tid == nrw || rp[tid + 1] > (u32)CSR_BATCH))
            *sh_nr = tid;
        if (tid == 0 && nrw == CSR_RB && rp[CSR_RB] <= (u32)CSR_BATCH) *sh_nr = CSR_RB;
        __syncthreads();
        const int nr = *sh_nr;
        if (nr == 0) {
            // the row alone holds more than a batch: total by a workgroup reduction, then its
            // entries in batch-sized chunks (not pipelined)
            const u32 len = rp[1];
            if (tid == 0) rowsum[0] = 0;
            __syncthreads();
            u64 part = 0;
            for (u32 q = tid; q < len; q += CSR_SBLOCK) {
                const int x = vslab[off + q];
                part += (u64)(x > 0 ? x : 0);
            }
            part = dme_wave_sum_u64(part);
            if ((tid & 63) == 0 && part) atomicAdd(&rowsum[0], part);
            __syncthreads();
            const u64 t = rowsum[0];
            if (tid == 0) count_total(t);
            const u32 tn = t > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (u32)t;
            for (u32 q0 = 0; q0 < len; q0 += CSR_BATCH) {
                const u32 nb = len - q0 < (u32)CSR_BATCH ? len - q0 : (u32)CSR_BATCH;
                fetch_entries(off + q0);
#pragma unroll
                for (int j = 0; j < CSR_EPT; ++j) n32[j] = tn;
                classify(nb);
                emit(c, v, n32);
            }
            r += 1;
            off += len;
            pre0 = fetch_ptr(r, tid);
            pre1 = tid == 0 ? fetch_ptr(r, CSR_SBLOCK) : 0;
            fetch_entries(off);
            continue;
        }
        const u32 nb = rp[nr];
        // the next batch's row pointers go out now; they are needed one trip later
        pre0 = fetch_ptr(r + nr, tid);
        pre1 = tid == 0 ? fetch_ptr(r + nr, CSR_SBLOCK) : 0;
        // Row totals and the row of every entry from two inclusive prefix sums over the batch (entry
        // order): S over the counts (stage area), total(t) = S[rp[t+1]-1] - S[rp[t]-1]; and C over
        // the number of rows starting at each position (16-bit, in the bkt area), row(i) = C[i].
        // Adding each entry to its row's sum with LDS atomics serialises (a wave's 64 consecutive
        // entries share one or two rows), and a binary search per entry costs ~50 VALU operations.
        u32* cntw = (u32*)bkt;  // two 16-bit counters per word
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            const u32 i = tid + j * CSR_SBLOCK;
            stage[i] = (i < nb && v[j] > 0) ? (u64)v[j] : 0;
        }
        for (int i = tid; i < CSR_BATCH / 2; i += CSR_SBLOCK) cntw[i] = 0;
        __syncthreads();
        if (tid >= 1 && tid < nr) {
            const u32 q = rp[tid];
            if (q < nb) atomicAdd(&cntw[q >> 1], (q & 1) ? 0x10000u : 1u);
        }
        __syncthreads();
        {
            u64 x[CSR_EPT];
            u64 sum = 0;
            u32 cs[CSR_EPT];
            u32 csum = 0;
            const uint4 cw = *(const uint4*)(cntw + tid * (CSR_EPT / 2));  // CSR_EPT == 8
            const u32 cwv[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
            for (int j = 0; j < CSR_EPT; ++j) {
                sum += stage[tid * CSR_EPT + j];
                x[j] = sum;
                csum += (cwv[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                cs[j] = csum;
            }
            u64 inc = sum;
            u32 cinc = csum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u64 o = __shfl_up(inc, d);
                const u32 oc = __shfl_up(cinc, d);
                if ((tid & 63) >= d) {
                    inc += o;
                    cinc += oc;
                }
            }
            u64* wtot = (u64*)(sh_nr + 2);  // [CSR_SWAVES], 8-byte aligned behind sh_nr
            u32* wcnt = (u32*)(wtot + CSR_SWAVES);
            if ((tid & 63) == 63) {
                wtot[tid >> 6] = inc;
                wcnt[tid >> 6] = cinc;
            }
            __syncthreads();
            u64 before = inc - sum;
            u32 cbefore = cinc - csum;
#pragma unroll
            for (int w = 0; w < CSR_SWAVES; ++w)
                if (w < (tid >> 6)) {
                    before += wtot[w];
                    cbefore += wcnt[w];
                }
#pragma unroll
            for (int j = 0; j < CSR_EPT; ++j) stage[tid * CSR_EPT + j] = before + x[j];
            uint4 outw;
            outw.x = (cbefore + cs[0]) | ((cbefore + cs[1]) << 16);
            outw.y = (cbefore + cs[2]) | ((cbefore + cs[3]) << 16);
            outw.z = (cbefore + cs[4]) | ((cbefore + cs[5]) << 16);
            outw.w = (cbefore + cs[6]) | ((cbefore + cs[7]) << 16);
            *(uint4*)(cntw + tid * (CSR_EPT / 2)) = outw;
        }
        __syncthreads();
        if (tid < nr) {
            const u32 lo = rp[tid], hi = rp[tid + 1];
            const u64 t = (hi ? stage[hi - 1] : 0) - (lo ? stage[lo - 1] : 0);
            rowsum[tid] = t;
            count_total(t);
        }
        int row[CSR_EPT];
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            const u32 i = tid + j * CSR_SBLOCK;
            row[j] = i < nb ? bkt[i] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            const u64 t = rowsum[row[j]];
            n32[j] = t > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (u32)t;
        }
        classify(nb);
        // the records move to their own registers; c/v take the next batch's loads, which then
        // overlap the ordering and the write-out below
        int pk[CSR_EPT], bk[CSR_EPT];
#pragma unroll
        for (int j = 0; j < CSR_EPT; ++j) {
            pk[j] = c[j];
            bk[j] = v[j];
        }
        off += nb;
        fetch_entries(off);
        emit(pk, bk, n32);  // its barriers also order the rowsum reads before the next zeroing
        r += nr;
    }
    if (tid == 0) sh_stat[0] = (u64)(r1 - r0);
    if (bad) atomicAdd(&sh_stat[3], bad);
    if (big) atomicAdd(&sh_stat[1], big);
    if (maxv) atomicMax(&sh_stat[4], (u64)maxv);
    if (maxt) atomicMax(&sh_stat[5], maxt);
    __syncthreads();
    for (int i = tid; i < a.TBL; i += CSR_SBLOCK) {
        const u32 cn = tb[i];
        if (cn) atomicAdd(&a.histV[i], (u64)cn);
    }
    if (tid <= 3 && sh_stat[tid]) atomicAdd(&a.status[tid], sh_stat[tid]);
    if (tid >= 4 && tid <= 5 && sh_stat[tid]) atomicMax(&a.status[tid], sh_stat[tid]);
}

struct BucketArgs {
    const uint2* entries;
    const u64* start;  // [NB + 1]
    long long K, nbins;
    u64 capacity;
    int sh, NB, BL, W, AC;
    u64* hist;
    double* init_acc;
    const u64* status;
    char* images;           // [NB + W] LDS images: AC x CB fp64 accumulators, then CB x BL counters
    long long image_bytes;  // = 8*AC*CB + 4*CB*BL (a multiple of 16)
};

// Entries per piece of the bucket-major stream (even, so that pieces start 16-byte aligned).
__device__ __forceinline__ u64 csr_piece(u64 total, int W) {
    const u64 piece = (total + W - 1) / W;
    return (piece + 1) & ~1ULL;
}

__global__ __launch_bounds__(CSR_BLOCK) void csr_bucket_kernel(BucketArgs a) {
    extern __shared__ u64 lds64[];
    const int CB = 1 << a.sh;
    // A wave's 64 entries often repeat a column (the most frequent word of a bucket holds ~30 % of
    // its entries) and same-address LDS atomics serialise: the accumulators are kept in AC copies
    // picked by the low bits of the row total, which are unrelated to the column.
    double* acc = (double*)lds64;        // [AC][CB]
    u32* h = (u32*)(acc + a.AC * CB);    // [CB][BL]
    const u32 acmask = (u32)a.AC - 1;
    const int tid = threadIdx.x;
    const u64 total = a.start[a.NB];
    if (total > a.capacity || a.status[6] != 0) return;  // malformed rowptr: flagged, nothing counted
    const u64 piece = csr_piece(total, a.W);
    u64 pos = piece * blockIdx.x;
    u64 end = pos + piece;
    if (end > total) end = total;
    if (pos >= end) return;
    // first bucket whose run ends beyond pos
    int lo = 0, hi = a.NB - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.start[mid + 1] > pos) hi = mid; else lo = mid + 1;
    }
    int b = lo;
    const u32 cmask = (u32)CB - 1;
    const int cells = CB * a.BL;
    while (pos < end) {
        u64 seg_end = a.start[b + 1];
        if (seg_end > end) seg_end = end;
        if (seg_end > pos) {
            for (int i = tid; i < cells; i += CSR_BLOCK) h[i] = 0;
            for (int i = tid; i < a.AC * CB; i += CSR_BLOCK) acc[i] = 0.0;
            __syncthreads();
            const long long kbase = (long long)b << a.sh;
            auto one = [&](u64 raw) {
                const u32 ex = (u32)raw, ey = (u32)(raw >> 32);
                const u32 v = ex >> a.sh;
                if (v == 0) return;
                const u32 cl = ex & cmask;
                if (v <= (u32)a.BL) atomicAdd(&h[cl * a.BL + (v - 1)], 1u);
                else atomicAdd(&a.hist[(kbase + cl) * a.nbins + v], 1ULL);
                const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)ey));
                __hip_atomic_fetch_add(&acc[(ey & acmask) * CB + cl], __dmul_rn((double)v, w),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            const u64* src = (const u64*)a.entries;
            u64 e = pos + tid;
            for (; e + 3ULL * CSR_BLOCK < seg_end; e += 4ULL * CSR_BLOCK) {  // four loads in flight
                const u64 x0 = __builtin_nontemporal_load(src + e);
                const u64 x1 = __builtin_nontemporal_load(src + e + CSR_BLOCK);
                const u64 x2 = __builtin_nontemporal_load(src + e + 2 * CSR_BLOCK);
                const u64 x3 = __builtin_nontemporal_load(src + e + 3 * CSR_BLOCK);
                one(x0);
                one(x1);
                one(x2);
                one(x3);
            }
            for (; e < seg_end; e += CSR_BLOCK) one(__builtin_nontemporal_load(src + e));
            __syncthreads();
            // The run (b, piece) is the (b + piece)-th of the stream: runs are ordered and each one
            // advances b or the piece.  Its LDS image goes out as it is; csr_reduce_kernel adds
            // the images of a bucket in a fixed order, so no global atomics are needed here.
            {
                uint4* dst = (uint4*)(a.images + (long long)(b + (int)blockIdx.x) * a.image_bytes);
                const uint4* src = (const uint4*)lds64;
                const int n16 = (int)(a.image_bytes >> 4);
                for (int i = tid; i < n16; i += CSR_BLOCK) dst[i] = src[i];
            }
            __syncthreads();
        }
        pos = seg_end;
        ++b;
    }
}

// hist[k][1..BL] += the counters of every run of bucket b; init_acc[k] += its accumulators.
// grid (NB, ceil(CB*BL / 1024)): the cells of a bucket are split over workgroups so that a bucket
// holding most of the stream (Zipf head in one bucket) is still summed in parallel.
__global__ __launch_bounds__(CSR_BLOCK) void csr_reduce_kernel(BucketArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const u64 s0 = a.start[b], s1 = a.start[b + 1];
    if (s1 == s0 || a.start[a.NB] > a.capacity || a.status[6] != 0) return;
    const int CB = 1 << a.sh;
    const u64 piece = csr_piece(a.start[a.NB], a.W);
    const int w_lo = (int)(s0 / piece), w_hi = (int)((s1 - 1) / piece);
    const long long kbase = (long long)b << a.sh;
    const int cells = CB * a.BL;
    const int i = blockIdx.y * CSR_BLOCK + tid;
    if (i < cells) {
        u64 sum = 0;
        const char* img = a.images + 8LL * a.AC * CB + 4LL * i;
        int w = w_lo;
        for (; w + 7 <= w_hi; w += 8) {  // eight loads in flight: a Zipf head bucket has hundreds of runs
            u32 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = *(const u32*)(img + (long long)(b + w + q) * a.image_bytes);
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += x[q];
        }
        for (; w <= w_hi; ++w) sum += *(const u32*)(img + (long long)(b + w) * a.image_bytes);
        if (sum) {
            const int cl = i / a.BL, bin = i - cl * a.BL + 1;
            a.hist[(kbase + cl) * a.nbins + bin] += sum;
        }
    }
    // the fp64 accumulators: block y takes the runs w = w_lo + y, + gridDim.y, ... (a Zipf head bucket
    // has hundreds of runs) and adds its partial sum with one atomic per column
    for (int c = tid; c < CB; c += CSR_BLOCK) {
        double sum = 0.0;
        for (int w = w_lo + (int)blockIdx.y; w <= w_hi; w += (int)gridDim.y) {
            const double* img = (const double*)(a.images + (long long)(b + w) * a.image_bytes);
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = q < a.AC ? img[q * CB + c] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum = __dadd_rn(sum, x[q]);
        }
        if (sum != 0.0) atomicAdd(&a.init_acc[kbase + c], sum);
    }
}

}  // namespace

extern "C" int64_t dme_stats_csr_workspace_bytes(int64_t N, int64_t nnz, int64_t K) {
    if (N <= 0 || K <= 0 || nnz < CSR_SMALL_NNZ || nnz >= (1LL << 32)) return 0;
    const long long b2 = dme_csr2_workspace_bytes(N, nnz, K);
    if (b2 > 0) return b2;  // K <= 65 536: the second design (dme_csr2.hip)
    CsrPlan p;
    if (plan_csr(N, nnz, K, 2, &p) != DME_OK) return 0;
    return p.bytes;
}

extern "C" int dme_stats_csr_i32(const int64_t* rowptr, const int32_t* colidx, const int32_t* vals,
                                 int64_t N, int64_t nnz, int64_t K, uint64_t* hist, int64_t nbins,
                                 uint64_t* histV, int64_t tbins, double* init_acc, uint64_t* status,
                                 uint64_t* overflow, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && nnz >= 0, "bad shape N=%lld K=%lld nnz=%lld", (long long)N,
                (long long)K, (long long)nnz);
    DME_REQUIRE(nbins >= 1 && tbins >= 1, "nbins/tbins must be >= 1");
    DME_REQUIRE(hist && histV && init_acc && status, "null output pointer");
    if (N == 0) return DME_OK;
    DME_REQUIRE(rowptr && (nnz == 0 || (colidx && vals)), "null CSR pointer");
    hipStream_t s = (hipStream_t)stream;

    CsrPlan p;
    // the scatter kernel addresses a slab's entries with 32-bit offsets
    bool partitioned = workspace != nullptr && nnz >= CSR_SMALL_NNZ && nnz < (1LL << 32);
    if (partitioned) {
        const int rc2 = dme_csr2_run((const long long*)rowptr, colidx, vals, N, nnz, K, (u64*)hist, nbins,
                                     (u64*)histV, tbins, init_acc, (u64*)status, (u64*)overflow,
                                     workspace, workspace_bytes, s);
        if (rc2 >= 0) return rc2;  // -1: more than 65 536 columns, the first design below
    }
    if (partitioned) {
        if (plan_csr(N, nnz, K, nbins, &p) != DME_OK) partitioned = false;  // K beyond the buckets
        else
            DME_REQUIRE(workspace_bytes >= p.bytes && ((uintptr_t)workspace & 255) == 0,
                        "CSR workspace: %lld bytes (256-byte aligned) needed, got %lld",
                        (long long)p.bytes, (long long)workspace_bytes);
    }
    if (!partitioned) {
        int cus = dme_device_cus();
        if (cus <= 0) cus = 256;
        long long waves = N < (long long)cus * 32 ? N : (long long)cus * 32;
        long long grid = (waves + 3) / 4;
        hipLaunchKernelGGL(stats_csr_kernel, dim3((unsigned)grid), dim3(256), 0, s,
                           (const long long*)rowptr, colidx, vals, (long long)N, (long long)nnz,
                           (long long)K, (u64*)hist, (long long)nbins, (u64*)histV, (long long)tbins,
                           init_acc, (u64*)status, (u64*)overflow);
        DME_HIP_CHECK(hipGetLastError());
        return DME_OK;
    }

    char* ws = (char*)workspace;
    u32* counts = (u32*)(ws + p.off_counts);
    u64* rel = (u64*)(ws + p.off_rel);
    u64* tot = (u64*)(ws + p.off_tot);
    u64* start = (u64*)(ws + p.off_start);
    uint2* entries = (uint2*)(ws + p.off_entries);

    hipLaunchKernelGGL(csr_count_kernel, dim3(p.G), dim3(CSR_BLOCK), (size_t)p.copies * p.NB * 4, s,
                       (const long long*)rowptr, colidx, (long long)N, (long long)nnz, (long long)K,
                       p.rows_per_slab, p.sh, p.NB, p.copies, counts, (u64*)status);
    hipLaunchKernelGGL(csr_offsets_kernel, dim3((p.NB + 63) / 64), dim3(CSR_BLOCK), 0, s, counts,
                       p.G, p.NB, rel, tot);
    hipLaunchKernelGGL(csr_bucket_scan_kernel, dim3(1), dim3(CSR_BLOCK), 0, s, tot, p.NB, start);

    CsrArgs a{};
    a.rowptr = (const long long*)rowptr;
    a.colidx = colidx;
    a.vals = vals;
    a.N = N; a.K = K; a.nbins = nbins; a.tbins = tbins;
    a.rows_per_slab = p.rows_per_slab;
    a.vlim = p.vlim;
    a.capacity = (u64)nnz;
    a.sh = p.sh; a.NB = p.NB;
    a.TBL = (int)(tbins < CSR_TBL_MAX ? tbins : CSR_TBL_MAX);
    a.rel = rel; a.start = start; a.entries = entries;
    a.hist = (u64*)hist; a.histV = (u64*)histV; a.init_acc = init_acc; a.status = (u64*)status;
    a.overflow = (u64*)overflow;
    const size_t lds_scatter = (size_t)scatter_lds(p.NB, a.TBL).bytes;
    DME_HIP_CHECK(hipFuncSetAttribute((const void*)csr_scatter_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(csr_scatter_kernel, dim3(p.G), dim3(CSR_SBLOCK), lds_scatter, s, a);

    BucketArgs bk{};
    bk.entries = entries; bk.start = start; bk.K = K; bk.nbins = nbins; bk.capacity = (u64)nnz;
    bk.sh = p.sh; bk.NB = p.NB; bk.BL = p.BL; bk.W = p.W; bk.AC = p.AC;
    bk.hist = (u64*)hist; bk.init_acc = init_acc; bk.status = (const u64*)status;
    const size_t lds_bucket = 8 * (size_t)p.AC * p.CB + 4 * (size_t)p.CB * p.BL;
    bk.images = ws + p.off_images;
    bk.image_bytes = (long long)lds_bucket;
    DME_HIP_CHECK(hipFuncSetAttribute((const void*)csr_bucket_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipLaunchKernelGGL(csr_bucket_kernel, dim3(p.W), dim3(CSR_BLOCK), lds_bucket, s, bk);
    hipLaunchKernelGGL(csr_reduce_kernel, dim3(p.NB, (p.CB * p.BL + CSR_BLOCK - 1) / CSR_BLOCK),
                       dim3(CSR_BLOCK), 0, s, bk);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
