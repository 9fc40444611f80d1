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
//   csr_scatter_kernel  wave per row: row total (DPP/shuffle reduction) -> histV in LDS; every entry
//                       is written as 8 bytes {column-in-bucket | count << cb, row total} to its
//                       bucket's run                                        [reads all, writes 8 B]
//   csr_bucket_kernel   the bucket-major entry stream is cut into equal pieces, one per
//                       workgroup; a piece holds one bucket (or a few): counts 1..BL of its CB
//                       columns are privatised in LDS (32-bit), the mean accumulator too (LDS fp64
//                       atomics); the LDS image of every (bucket, piece) run is dumped as it is
//   csr_reduce_kernel   adds the images of each bucket into hist / init_acc (plain +=: a bucket's
//                       columns belong to one workgroup row of the grid)
//
// Integer results are exact and order-independent; init_acc is summed in arrival order (fp64), so
// its last bits vary from run to run, as with the atomics path.  Without a workspace, or for inputs
// too small to repay five launches, the atomics kernel runs.
#include "dme_common.h"

// compile-time ablation for profiling experiments (0 in every shipped build)
#ifndef DME_CSR_ABLATE
#define DME_CSR_ABLATE 0
#endif

namespace {

constexpr int CSR_BLOCK = 1024;
constexpr int CSR_WAVES = CSR_BLOCK / 64;
constexpr int CSR_MAX_BUCKETS = 2048;
constexpr int CSR_HIST_LDS_BYTES = 60 * 1024;
constexpr int CSR_TBL_MAX = 8192;  // row totals below this are counted in LDS
constexpr long long CSR_SMALL_NNZ = 1 << 15;

// ---------------------------------------------------------------------------------------------
// atomics path (small inputs, no workspace)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_csr_kernel(const long long* rowptr, const int* colidx,
                                                        const int* vals, long long N, long long K,
                                                        u64* hist, long long nbins, u64* histV,
                                                        long long tbins, double* init_acc,
                                                        u64* status) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    __shared__ u64 sh_stat[8];  // [1] cells >= nbins [2] rows >= tbins [3] bad entries [4],[5] maxima
    if (threadIdx.x < 8) sh_stat[threadIdx.x] = 0;
    __syncthreads();
    u64 nrows = 0;
    for (long long i = wave0; i < N; i += nwaves) {
        const long long p0 = rowptr[i], p1 = rowptr[i + 1];
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
            else atomicAdd(&sh_stat[1], 1ULL);
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
    long long G = 2LL * cus;
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

// counts[g][b] = entries of slab g whose column lies in bucket b
__global__ __launch_bounds__(CSR_BLOCK) void csr_count_kernel(const long long* rowptr,
                                                              const int* colidx, long long N,
                                                              long long K, long long rows_per_slab,
                                                              int sh, int NB, int copies,
                                                              u32* counts) {
    extern __shared__ u32 cnt[];  // [copies][NB]
    const int tid = threadIdx.x;
    for (int i = tid; i < copies * NB; i += CSR_BLOCK) cnt[i] = 0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > N) r1 = N;
    if (r0 < r1) {
        const long long e0 = rowptr[r0], e1 = rowptr[r1];
        u32* mine = cnt + ((tid >> 6) & (copies - 1)) * NB;
        long long e = e0 + tid;
        for (; e + 3LL * CSR_BLOCK < e1; e += 4LL * CSR_BLOCK) {
            const int c0 = __builtin_nontemporal_load(colidx + e);
            const int c1 = __builtin_nontemporal_load(colidx + e + CSR_BLOCK);
            const int c2 = __builtin_nontemporal_load(colidx + e + 2 * CSR_BLOCK);
            const int c3 = __builtin_nontemporal_load(colidx + e + 3 * CSR_BLOCK);
            if ((u64)(long long)c0 < (u64)K) atomicAdd(&mine[c0 >> sh], 1u);
            if ((u64)(long long)c1 < (u64)K) atomicAdd(&mine[c1 >> sh], 1u);
            if ((u64)(long long)c2 < (u64)K) atomicAdd(&mine[c2 >> sh], 1u);
            if ((u64)(long long)c3 < (u64)K) atomicAdd(&mine[c3 >> sh], 1u);
        }
        for (; e < e1; e += CSR_BLOCK) {
            const int c = colidx[e];
            if ((u64)(long long)c < (u64)K) atomicAdd(&mine[c >> sh], 1u);
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
    int sh, NB, TBL;
    const u64* rel;
    const u64* start;
    uint2* entries;
    u64* hist;
    u64* histV;
    double* init_acc;
    u64* status;
};

__global__ __launch_bounds__(CSR_BLOCK) void csr_scatter_kernel(CsrArgs a) {
    extern __shared__ u64 lds64[];
    // [NB] run bases (u64) | [NB] cursors (u32) | [TBL] row-total bins (u32) | [8] status
    u64* base = lds64;
    u32* cursor = (u32*)(base + a.NB);
    u32* tb = cursor + a.NB + (a.NB & 1);
    u64* sh_stat = (u64*)(tb + a.TBL + (a.TBL & 1));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long g = blockIdx.x;
    for (int b = tid; b < a.NB; b += CSR_BLOCK) {
        base[b] = a.start[b] + a.rel[g * a.NB + b];
        cursor[b] = 0;
    }
    for (int i = tid; i < a.TBL; i += CSR_BLOCK) tb[i] = 0;
    if (tid < 8) sh_stat[tid] = 0;
    __syncthreads();
    const long long r0 = g * a.rows_per_slab;
    long long r1 = r0 + a.rows_per_slab;
    if (r1 > a.N) r1 = a.N;
    const u32 cmask = (1u << a.sh) - 1;
    u32 maxv = 0;
    u64 maxt = 0, bad = 0, big = 0, nrows = 0;

    // one stored entry: validate, pack, take a slot in its bucket's run, store
    auto put = [&](long long k, int v, u64 total, u32 n32) {
        if ((u64)k >= (u64)a.K) {  // not counted by csr_count_kernel either
            if (v != 0) bad++;
            return;
        }
        u32 packed = 0;  // count 0: the bucket pass skips it
        if (v < 0) {
            bad++;
        } else if (v > 0) {
            maxv = (u32)v > maxv ? (u32)v : maxv;
            if ((long long)v < a.vlim) {
                packed = ((u32)k & cmask) | ((u32)v << a.sh);
            } else {
                if ((long long)v < a.nbins) atomicAdd(&a.hist[k * a.nbins + v], 1ULL);
                else big++;
                const double w = __ddiv_rn(1.0, __dadd_rn(1.0, (double)total));
                atomicAdd(&a.init_acc[k], __dmul_rn((double)v, w));
            }
        }
        const int b = (int)(k >> a.sh);
        const u32 slot = (DME_CSR_ABLATE & 16) ? (u32)lane : atomicAdd(&cursor[b], 1u);
        if (!(DME_CSR_ABLATE & 8)) a.entries[base[b] + slot] = make_uint2(packed, n32);
        else if (packed == 0xFFFFFFFFu) a.entries[slot] = make_uint2(packed, n32);
    };

    // Each wave takes R consecutive rows per trip, and the trips are software-pipelined: while trip
    // t is processed, the entries of trip t+1 and the row pointers of trip t+2 are in flight.
    // With one row at a time the pass is bound by the rowptr -> entries -> store latency chain.
    constexpr int R = 4;
    const long long stride = (long long)CSR_WAVES * R;
    auto load_ptrs = [&](long long r) -> long long {  // rowptr[r .. r+R] on lanes 0..R
        long long rr = r + lane;
        if (rr > r1) rr = r1;
        return (lane <= R && r < r1) ? a.rowptr[rr] : 0;
    };
    long long r = r0 + (long long)wave * R;
    long long pl_nxt = load_ptrs(r);
    long long pl_n2 = load_ptrs(r + stride);
    long long p[R + 1];
    int c[R], v[R];
#pragma unroll
    for (int j = 0; j <= R; ++j) p[j] = __shfl(pl_nxt, j);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long long q = p[j] + lane;
        const bool in = q < p[j + 1];
        c[j] = in ? __builtin_nontemporal_load(a.colidx + q) : 0;
        v[j] = in ? __builtin_nontemporal_load(a.vals + q) : 0;
    }
    pl_nxt = pl_n2;
    for (; r < r1; r += stride) {
        pl_n2 = load_ptrs(r + 2 * stride);
        long long pn[R + 1];
        int cn[R], vn[R];
#pragma unroll
        for (int j = 0; j <= R; ++j) pn[j] = __shfl(pl_nxt, j);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const long long q = pn[j] + lane;
            const bool in = q < pn[j + 1];
            cn[j] = in ? __builtin_nontemporal_load(a.colidx + q) : 0;
            vn[j] = in ? __builtin_nontemporal_load(a.vals + q) : 0;
        }
        u64 total[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            u64 sacc = (u64)(v[j] > 0 ? v[j] : 0);
            for (long long q = p[j] + 64 + lane; q < p[j + 1]; q += 64) {  // rows beyond 64 entries
                const int x = a.vals[q];
                sacc += (u64)(x > 0 ? x : 0);
            }
            if (sacc > 0xFFFFFFFFFFFFULL) sacc = 0xFFFFFFFFFFFFULL;
            total[j] = (u64)dme_wave_sum_u32((u32)(sacc & 0xFFFFFF)) +
                       ((u64)dme_wave_sum_u32((u32)(sacc >> 24)) << 24);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (r + j >= r1) break;
            const u64 t = total[j];
            nrows++;
            if (lane == 0 && t > 0) {
                maxt = t > maxt ? t : maxt;
                if (t < (u64)a.TBL) atomicAdd(&tb[t], 1u);
                else if ((long long)t < a.tbins) atomicAdd(&a.histV[t], 1ULL);
                else atomicAdd(&sh_stat[2], 1ULL);
            }
            const u32 n32 = t > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (u32)t;
            if (p[j] + lane < p[j + 1]) put(c[j], v[j], t, n32);
            for (long long q = p[j] + 64 + lane; q < p[j + 1]; q += 64)
                put(a.colidx[q], a.vals[q], t, n32);
        }
#pragma unroll
        for (int j = 0; j <= R; ++j) p[j] = pn[j];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            c[j] = cn[j];
            v[j] = vn[j];
        }
        pl_nxt = pl_n2;
    }
    if (bad) atomicAdd(&sh_stat[3], bad);
    if (big) atomicAdd(&sh_stat[1], big);
    if (maxv) atomicMax(&sh_stat[4], (u64)maxv);
    if (maxt) atomicMax(&sh_stat[5], maxt);
    if (lane == 0 && nrows) atomicAdd(&sh_stat[0], nrows);
    __syncthreads();
    for (int i = tid; i < a.TBL; i += CSR_BLOCK) {
        const u32 c = tb[i];
        if (c) atomicAdd(&a.histV[i], (u64)c);
    }
    if (tid <= 3 && sh_stat[tid]) atomicAdd(&a.status[tid], sh_stat[tid]);
    if (tid >= 4 && tid <= 5 && sh_stat[tid]) atomicMax(&a.status[tid], sh_stat[tid]);
}

struct BucketArgs {
    const uint2* entries;
    const u64* start;  // [NB + 1]
    long long K, nbins;
    int sh, NB, BL, W, AC;
    u64* hist;
    double* init_acc;
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
                const u32 cl = (DME_CSR_ABLATE & 128) ? ((ex + (u32)tid) & cmask) : (ex & cmask);
                if (!(DME_CSR_ABLATE & 1)) {
                    if (v <= (u32)a.BL) atomicAdd(&h[cl * a.BL + (v - 1)], 1u);
                    else atomicAdd(&a.hist[(kbase + cl) * a.nbins + v], 1ULL);
                }
                const double w = (DME_CSR_ABLATE & 4) ? (double)ey
                                                      : __ddiv_rn(1.0, __dadd_rn(1.0, (double)ey));
                if (DME_CSR_ABLATE & 256)
                    atomicAdd((u64*)&acc[(ey & acmask) * CB + cl], (u64)(__dmul_rn((double)v, w) * 1e12));
                else if (DME_CSR_ABLATE & 512)
                    atomicAdd((float*)&acc[(ey & acmask) * CB + cl], (float)__dmul_rn((double)v, w));
                else if (!(DME_CSR_ABLATE & 2))
                    __hip_atomic_fetch_add(&acc[(ey & acmask) * CB + cl], __dmul_rn((double)v, w),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (w == 12345.0) acc[cl] = w;
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
    if (s1 == s0) return;
    const int CB = 1 << a.sh;
    const u64 piece = csr_piece(a.start[a.NB], a.W);
    const int w_lo = (int)(s0 / piece), w_hi = (int)((s1 - 1) / piece);
    const long long kbase = (long long)b << a.sh;
    const int cells = CB * a.BL;
    const int i = blockIdx.y * CSR_BLOCK + tid;
    if (i < cells) {
        u64 sum = 0;
        for (int w = w_lo; w <= w_hi; ++w)
            sum += ((const u32*)(a.images + (long long)(b + w) * a.image_bytes + 8LL * a.AC * CB))[i];
        if (sum) {
            const int cl = i / a.BL, bin = i - cl * a.BL + 1;
            a.hist[(kbase + cl) * a.nbins + bin] += sum;
        }
    }
    if (blockIdx.y == 0) {
        for (int c = tid; c < CB; c += CSR_BLOCK) {
            double sum = 0.0;
            for (int w = w_lo; w <= w_hi; ++w) {
                const double* img = (const double*)(a.images + (long long)(b + w) * a.image_bytes);
                for (int q = 0; q < a.AC; ++q) sum = __dadd_rn(sum, img[q * CB + c]);
            }
            if (sum != 0.0) a.init_acc[kbase + c] = __dadd_rn(a.init_acc[kbase + c], sum);
        }
    }
}

}  // namespace

extern "C" int64_t dme_stats_csr_workspace_bytes(int64_t N, int64_t nnz, int64_t K) {
    if (N <= 0 || K <= 0 || nnz < CSR_SMALL_NNZ) return 0;
    CsrPlan p;
    if (plan_csr(N, nnz, K, 2, &p) != DME_OK) return 0;
    return p.bytes;
}

extern "C" int dme_stats_csr_i32(const int64_t* rowptr, const int32_t* colidx, const int32_t* vals,
                                 int64_t N, int64_t nnz, int64_t K, uint64_t* hist, int64_t nbins,
                                 uint64_t* histV, int64_t tbins, double* init_acc, uint64_t* status,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && nnz >= 0, "bad shape N=%lld K=%lld nnz=%lld", (long long)N,
                (long long)K, (long long)nnz);
    DME_REQUIRE(nbins >= 1 && tbins >= 1, "nbins/tbins must be >= 1");
    DME_REQUIRE(hist && histV && init_acc && status, "null output pointer");
    if (N == 0) return DME_OK;
    DME_REQUIRE(rowptr && (nnz == 0 || (colidx && vals)), "null CSR pointer");
    hipStream_t s = (hipStream_t)stream;

    CsrPlan p;
    bool partitioned = workspace != nullptr && nnz >= CSR_SMALL_NNZ;
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
                           (const long long*)rowptr, colidx, vals, (long long)N, (long long)K,
                           (u64*)hist, (long long)nbins, (u64*)histV, (long long)tbins, init_acc,
                           (u64*)status);
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
                       (const long long*)rowptr, colidx, (long long)N, (long long)K,
                       p.rows_per_slab, p.sh, p.NB, p.copies, counts);
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
    a.sh = p.sh; a.NB = p.NB;
    a.TBL = (int)(tbins < CSR_TBL_MAX ? tbins : CSR_TBL_MAX);
    a.rel = rel; a.start = start; a.entries = entries;
    a.hist = (u64*)hist; a.histV = (u64*)histV; a.init_acc = init_acc; a.status = (u64*)status;
    const size_t lds_scatter = 8 * (size_t)p.NB + 4 * (size_t)(p.NB + (p.NB & 1)) +
                               4 * (size_t)(a.TBL + (a.TBL & 1)) + 64;
    if (!(DME_CSR_ABLATE & 64))
        hipLaunchKernelGGL(csr_scatter_kernel, dim3(p.G), dim3(CSR_BLOCK), lds_scatter, s, a);

    BucketArgs bk{};
    bk.entries = entries; bk.start = start; bk.K = K; bk.nbins = nbins;
    bk.sh = p.sh; bk.NB = p.NB; bk.BL = p.BL; bk.W = p.W; bk.AC = p.AC;
    bk.hist = (u64*)hist; bk.init_acc = init_acc;
    const size_t lds_bucket = 8 * (size_t)p.AC * p.CB + 4 * (size_t)p.CB * p.BL;
    bk.images = ws + p.off_images;
    bk.image_bytes = (long long)lds_bucket;
    DME_HIP_CHECK(hipFuncSetAttribute((const void*)csr_bucket_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    if (!(DME_CSR_ABLATE & 32))
        hipLaunchKernelGGL(csr_bucket_kernel, dim3(p.W), dim3(CSR_BLOCK), lds_bucket, s, bk);
    hipLaunchKernelGGL(csr_reduce_kernel, dim3(p.NB, (p.CB * p.BL + CSR_BLOCK - 1) / CSR_BLOCK),
                       dim3(CSR_BLOCK), 0, s, bk);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
