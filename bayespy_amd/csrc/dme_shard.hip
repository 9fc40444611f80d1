// dme_shard.hip — message of the multi-GPU statistics sum (rows shard over GPUs, histograms add).
//
// The reference is one process (no counterpart); what is summed are the partial results of
// CompressedRowData.appendRow over disjoint row sets (DE/dme.py:116-123) and of the CLI's mean
// accumulator (DE/fdp.py:87-89).  ONE integer SUM all-reduce (RCCL over xGMI) carries
//     hist | histV | status[8] | slots[world][2 + K]
// where a rank fills only its own slot row: its two observed maxima (status[4], status[5], which
// do not add) and the bit patterns of its K fp64 mean accumulators.  The sum leaves every rank's
// slot row unchanged, so afterwards every rank takes the maxima and adds the accumulators in rank
// order: the start point of the replicated solve is bit-identical on every rank and from run to
// run.  These two kernels write and read that message; the collective itself is torch.distributed's.
#include "dme_common.h"

namespace {

constexpr int SHARD_BLOCK = 256;

__global__ __launch_bounds__(SHARD_BLOCK) void shard_pack_kernel(
    const u64* hist, long long nh, const u64* histV, long long nv, const u64* status,
    const double* init_acc, long long K, long long rank, long long world, u64* msg) {
    const long long total = nh + nv + 8 + world * (2 + K);
    const long long stride = (long long)gridDim.x * SHARD_BLOCK;
    for (long long i = (long long)blockIdx.x * SHARD_BLOCK + threadIdx.x; i < total; i += stride) {
        u64 x;
        if (i < nh) {
            x = hist[i];
        } else if (i < nh + nv) {
            x = histV[i - nh];
        } else if (i < nh + nv + 8) {
            const long long j = i - nh - nv;
            x = (j == 4 || j == 5) ? 0ULL : status[j];  // the maxima travel in the slots
        } else {
            const long long q = i - nh - nv - 8;
            const long long r = q / (2 + K), j = q - r * (2 + K);
            if (r != rank) x = 0ULL;
            else if (j < 2) x = status[4 + j];
            else x = (u64)__double_as_longlong(init_acc[j - 2]);
        }
        msg[i] = x;
    }
}

__global__ __launch_bounds__(SHARD_BLOCK) void shard_unpack_kernel(u64* msg, long long nh,
                                                                   long long nv, long long K,
                                                                   long long world,
                                                                   double* init_acc) {
    u64* status = msg + nh + nv;
    const u64* slots = status + 8;
    const long long t = (long long)blockIdx.x * SHARD_BLOCK + threadIdx.x;
    if (t < 2) {
        u64 m = 0;
        for (long long r = 0; r < world; ++r) {
            const u64 x = slots[r * (2 + K) + t];
            m = x > m ? x : m;
        }
        status[4 + t] = m;
    }
    if (t < K) {
        double s = __longlong_as_double((long long)slots[2 + t]);
        for (long long r = 1; r < world; ++r)  // rank order: the same bits on every rank
            s = __dadd_rn(s, __longlong_as_double((long long)slots[r * (2 + K) + 2 + t]));
        init_acc[t] = s;
    }
}

// The same message in 32-bit words, for jobs of fewer than 2^32 rows (no bin can hold more than the
// job's rows): hist and histV take one word per bin.  Everything else is 64-bit and travels as (lo, hi)
// word pairs, which a 32-bit SUM can only transport, not add (a carry out of the low word is lost): so a
// rank's eight status words ride in its OWN slot row, in front of its mean accumulators,
//     msg32 = hist[nh] | histV[nv] | slots[world][8 + K] as (lo, hi)
// every other rank's row is 0, and the sum hands every rank all rows unchanged.  unpack32 adds the
// additive status words in 64 bits (exact whatever the shards' sizes) and takes the maxima.
__global__ __launch_bounds__(SHARD_BLOCK) void shard_pack32_kernel(
    const u64* hist, long long nh, const u64* histV, long long nv, const u64* status,
    const double* init_acc, long long K, long long rank, long long world, u32* msg) {
    const long long nb = nh + nv;
    const long long total = nb + 2 * world * (8 + K);
    const long long stride = (long long)gridDim.x * SHARD_BLOCK;
    for (long long i = (long long)blockIdx.x * SHARD_BLOCK + threadIdx.x; i < total; i += stride) {
        u32 x;
        if (i < nh) {
            x = (u32)hist[i];
        } else if (i < nb) {
            x = (u32)histV[i - nh];
        } else {
            const long long q = (i - nb) >> 1;  // 64-bit word index behind the bins
            const int half = (int)((i - nb) & 1);
            const long long r = q / (8 + K), j = q - r * (8 + K);
            u64 v;
            if (r != rank) v = 0ULL;
            else if (j < 8) v = status[j];
            else v = (u64)__double_as_longlong(init_acc[j - 8]);
            x = half ? (u32)(v >> 32) : (u32)v;
        }
        msg[i] = x;
    }
}

// After the all-reduce: widens the bins into hist64 / histV64 (what dme_suffix_scan reads), rebuilds
// status[8] (sums and maxima over the ranks' slot rows) and adds the mean accumulators in rank order.
__global__ __launch_bounds__(SHARD_BLOCK) void shard_unpack32_kernel(
    const u32* msg, long long nh, long long nv, long long K, long long world, u64* hist64,
    u64* histV64, u64* status, double* init_acc) {
    const long long nb = nh + nv;
    const long long stride = (long long)gridDim.x * SHARD_BLOCK;
    const long long t0 = (long long)blockIdx.x * SHARD_BLOCK + threadIdx.x;
    for (long long i = t0; i < nb; i += stride) {
        const u64 x = (u64)msg[i];
        if (i < nh) hist64[i] = x;
        else histV64[i - nh] = x;
    }
    const u32* tail = msg + nb;
    auto word = [&](long long w) { return (u64)tail[2 * w] | ((u64)tail[2 * w + 1] << 32); };
    if (t0 < 8) {
        u64 v = 0;
        for (long long r = 0; r < world; ++r) {
            const u64 x = word(r * (8 + K) + t0);
            if (t0 == 4 || t0 == 5) v = x > v ? x : v;  // the observed maxima
            else v += x;
        }
        status[t0] = v;
    }
    for (long long k = t0; k < K; k += stride) {
        double s = __longlong_as_double((long long)word(8 + k));
        for (long long r = 1; r < world; ++r)  // rank order: the same bits on every rank
            s = __dadd_rn(s, __longlong_as_double((long long)word(r * (8 + K) + 8 + k)));
        init_acc[k] = s;
    }
}

// ---- compact form of the count histograms for large K --------------------------------------------
// hist is K x nbins with nbins set by the largest count of ANY column; with many columns (bag of
// words: K = 50 000) nearly all of them stop far below it.  The bins 1 .. ext[k] of every column,
// laid end to end, are what needs to travel: ext[k] = the column's largest occupied bin, agreed
// over the ranks by a MAX all-reduce; the offsets are its prefix sums.  dme_shard_pack(32) then takes
// the compact array in place of hist.
constexpr int COLS_PER_BLOCK = SHARD_BLOCK / 64;  // a wave per column

__global__ __launch_bounds__(SHARD_BLOCK) void shard_extents_kernel(const u64* hist, long long K,
                                                                    long long nbins, long long* ext) {
    const int lane = threadIdx.x & 63;
    for (long long k = (long long)blockIdx.x * COLS_PER_BLOCK + (threadIdx.x >> 6); k < K;
         k += (long long)gridDim.x * COLS_PER_BLOCK) {
        const u64* row = hist + k * nbins;
        u64 m = 0;
        for (long long v = 1 + lane; v < nbins; v += 64)
            if (row[v]) m = (u64)v;  // v grows: the last hit is the lane's largest
        m = dme_wave_max_u64(m);
        if (lane == 0) ext[k] = (long long)m;
    }
}

// off[k] = ext[0] + ... + ext[k-1], off[K] = the compact length.  One workgroup: a contiguous run of
// columns per thread, the threads' sums scanned in LDS.
__global__ __launch_bounds__(1024) void shard_offsets_kernel(const long long* ext, long long K,
                                                             long long* off) {
    __shared__ long long sh[1024];
    const long long per = (K + 1023) / 1024;
    const long long lo = (long long)threadIdx.x * per, hi = lo + per < K ? lo + per : K;
    long long s = 0;
    for (long long k = lo; k < hi; ++k) s += ext[k] > 0 ? ext[k] : 0;
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = sh[threadIdx.x] - s;  // exclusive
    for (long long k = lo; k < hi; ++k) {
        off[k] = run;
        run += ext[k] > 0 ? ext[k] : 0;
    }
    if (threadIdx.x == 1023) off[K] = sh[1023];
}

// TO_COMPACT: compact[off[k] + v - 1] = hist[k][v] for v in 1 .. off[k+1] - off[k] (0 beyond the dense
// bins: those counts come from the overflow store), the words from off[K] to `capacity` are zeroed, and
// a compact length beyond `capacity` raises status[7] (nothing is written past it).
// Otherwise the way back: hist[k][v] = that word, and 0 in every other bin of the row; cells beyond the
// dense bins go back to the overflow store (status[1] counts those it cannot hold).
template <bool TO_COMPACT>
__global__ __launch_bounds__(SHARD_BLOCK) void shard_compact_kernel(u64* hist, long long K,
                                                                    long long nbins,
                                                                    const long long* off,
                                                                    u64* compact, long long capacity,
                                                                    u64* overflow, u64* status) {
    const int lane = threadIdx.x & 63;
    const long long total = off[K];
    if (TO_COMPACT) {
        const long long t = (long long)blockIdx.x * SHARD_BLOCK + threadIdx.x;
        if (t == 0 && total > capacity && status) atomicAdd(&status[7], 1ULL);
        for (long long i = total + t; i < capacity; i += (long long)gridDim.x * SHARD_BLOCK) compact[i] = 0ULL;
    }
    for (long long k = (long long)blockIdx.x * COLS_PER_BLOCK + (threadIdx.x >> 6); k < K;
         k += (long long)gridDim.x * COLS_PER_BLOCK) {
        const long long o = off[k];
        const long long n = off[k + 1] - o;  // agreed over the ranks: may exceed this rank's bins
        u64* row = hist + k * nbins;
        if constexpr (TO_COMPACT) {
            for (long long v = 1 + lane; v <= n; v += 64)
                if (o + v - 1 < capacity) compact[o + v - 1] = v < nbins ? row[v] : 0ULL;
        } else {
            for (long long v = lane; v < nbins; v += 64)
                row[v] = (v >= 1 && v <= n && o + v - 1 < capacity) ? compact[o + v - 1] : 0ULL;
            for (long long v = nbins + lane; v <= n; v += 64) {
                const u64 c = o + v - 1 < capacity ? compact[o + v - 1] : 0ULL;
                if (c && !dme_overflow_put(overflow, (u64)k, (u64)v, c) && status) atomicAdd(&status[1], c);
            }
        }
    }
}

// ext[k] = max(ext[k], v) over the overflow records (the dense bins' extents are there already)
__global__ __launch_bounds__(SHARD_BLOCK) void overflow_extents_kernel(const u64* overflow, long long K,
                                                                       long long* ext) {
    const u64 used = overflow[0] < overflow[1] ? overflow[0] : overflow[1];
    const DmeOvfRec* rec = (const DmeOvfRec*)(overflow + 2);
    for (u64 i = (u64)blockIdx.x * SHARD_BLOCK + threadIdx.x; i < used; i += (u64)gridDim.x * SHARD_BLOCK) {
        const DmeOvfRec r = rec[i];
        if (r.n && (long long)r.k < K) atomicMax((unsigned long long*)&ext[r.k], (unsigned long long)r.v);
    }
}

// compact[off[k] + v - 1] += n for every record (after shard_compact_kernel<true> has written the dense part)
__global__ __launch_bounds__(SHARD_BLOCK) void overflow_scatter_kernel(const u64* overflow, long long K,
                                                                       const long long* off, u64* compact,
                                                                       long long capacity) {
    const u64 used = overflow[0] < overflow[1] ? overflow[0] : overflow[1];
    const DmeOvfRec* rec = (const DmeOvfRec*)(overflow + 2);
    for (u64 i = (u64)blockIdx.x * SHARD_BLOCK + threadIdx.x; i < used; i += (u64)gridDim.x * SHARD_BLOCK) {
        const DmeOvfRec r = rec[i];
        if (!r.n || (long long)r.k >= K || r.v == 0) continue;
        const long long o = off[r.k], n = off[r.k + 1] - o;
        if ((long long)r.v <= n && o + r.v - 1 < capacity) atomicAdd(&compact[o + r.v - 1], r.n);
    }
}

// dst gets src's records appended (merging two stores)
__global__ __launch_bounds__(SHARD_BLOCK) void overflow_append_kernel(u64* dst, const u64* src, u64* status) {
    const u64 used = src[0] < src[1] ? src[0] : src[1];
    const DmeOvfRec* rec = (const DmeOvfRec*)(src + 2);
    for (u64 i = (u64)blockIdx.x * SHARD_BLOCK + threadIdx.x; i < used; i += (u64)gridDim.x * SHARD_BLOCK) {
        const DmeOvfRec r = rec[i];
        if (r.n && !dme_overflow_put(dst, r.k, r.v, r.n) && status) atomicAdd(&status[1], r.n);
    }
}

__global__ void overflow_init_kernel(u64* store, u64 capacity) {
    store[0] = 0;
    store[1] = capacity;
}

// The ragged layout's two offset tables in one launch (one workgroup, as shard_offsets_kernel):
// off[k] = ext[0] + ... + ext[k-1] and slotoff[k] = the same over ceil(ext / seg), both with K + 1 entries.
__global__ __launch_bounds__(1024) void ragged_plan_kernel(const long long* ext, long long K, long long seg,
                                                           long long* off, long long* slotoff) {
    __shared__ long long sh[2][1024];
    const long long per = (K + 1023) / 1024;
    const long long lo = (long long)threadIdx.x * per, hi = lo + per < K ? lo + per : K;
    long long s = 0, q = 0;
    for (long long k = lo; k < hi; ++k) {
        const long long e = ext[k] > 0 ? ext[k] : 0;
        s += e;
        q += (e + seg - 1) / seg;
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long a0 = threadIdx.x >= d ? sh[0][threadIdx.x - d] : 0;
        const long long a1 = threadIdx.x >= d ? sh[1][threadIdx.x - d] : 0;
        __syncthreads();
        sh[0][threadIdx.x] += a0;
        sh[1][threadIdx.x] += a1;
        __syncthreads();
    }
    long long run = sh[0][threadIdx.x] - s, qrun = sh[1][threadIdx.x] - q;  // exclusive
    for (long long k = lo; k < hi; ++k) {
        const long long e = ext[k] > 0 ? ext[k] : 0;
        off[k] = run;
        slotoff[k] = qrun;
        run += e;
        qrun += (e + seg - 1) / seg;
    }
    if (threadIdx.x == 1023) {
        off[K] = sh[0][1023];
        slotoff[K] = sh[1][1023];
    }
}

long long overflow_grid(long long n) {
    long long g = (n + SHARD_BLOCK - 1) / SHARD_BLOCK;
    if (g > 1024) g = 1024;
    return g < 1 ? 1 : g;
}

long long shard_col_grid(long long K) {
    long long grid = (K + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
    if (grid > 8192) grid = 8192;
    return grid < 1 ? 1 : grid;
}

}  // namespace

extern "C" int dme_shard_extents(const uint64_t* hist, int64_t K, int64_t nbins, int64_t* ext,
                                 void* stream) {
    DME_REQUIRE(hist && ext && K >= 1 && nbins >= 1, "bad arguments");
    hipLaunchKernelGGL(shard_extents_kernel, dim3((unsigned)shard_col_grid(K)), dim3(SHARD_BLOCK), 0,
                       (hipStream_t)stream, (const u64*)hist, (long long)K, (long long)nbins,
                       (long long*)ext);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_shard_offsets(const int64_t* ext, int64_t K, int64_t* off, void* stream) {
    DME_REQUIRE(ext && off && K >= 1, "bad arguments");
    hipLaunchKernelGGL(shard_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       (const long long*)ext, (long long)K, (long long*)off);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_shard_compact(const uint64_t* hist, int64_t K, int64_t nbins,
                                 const uint64_t* overflow, const int64_t* off, uint64_t* compact,
                                 int64_t capacity, uint64_t* status, void* stream) {
    DME_REQUIRE(hist && off && compact && K >= 1 && nbins >= 1 && capacity >= 0, "bad arguments");
    hipLaunchKernelGGL(shard_compact_kernel<true>, dim3((unsigned)shard_col_grid(K)),
                       dim3(SHARD_BLOCK), 0, (hipStream_t)stream, (u64*)hist, (long long)K,
                       (long long)nbins, (const long long*)off, (u64*)compact, (long long)capacity,
                       (u64*)nullptr, (u64*)status);
    if (overflow)
        hipLaunchKernelGGL(overflow_scatter_kernel, dim3(256), dim3(SHARD_BLOCK), 0, (hipStream_t)stream,
                           (const u64*)overflow, (long long)K, (const long long*)off, (u64*)compact,
                           (long long)capacity);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_shard_expand(const uint64_t* compact, int64_t capacity, int64_t K, int64_t nbins,
                                const int64_t* off, uint64_t* hist, uint64_t* overflow,
                                uint64_t* status, void* stream) {
    DME_REQUIRE(hist && off && compact && K >= 1 && nbins >= 1 && capacity >= 0, "bad arguments");
    hipLaunchKernelGGL(shard_compact_kernel<false>, dim3((unsigned)shard_col_grid(K)),
                       dim3(SHARD_BLOCK), 0, (hipStream_t)stream, (u64*)hist, (long long)K,
                       (long long)nbins, (const long long*)off, (u64*)compact, (long long)capacity,
                       (u64*)overflow, (u64*)status);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int64_t dme_overflow_bytes(int64_t capacity) {
    if (capacity < 0 || capacity > (1LL << 40)) return -1;
    return 16 + 16 * capacity;
}

extern "C" int dme_overflow_init(uint64_t* store, int64_t capacity, void* stream) {
    DME_REQUIRE(store && capacity >= 0 && ((uintptr_t)store % 16) == 0, "bad overflow store");
    hipLaunchKernelGGL(overflow_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (u64*)store,
                       (u64)capacity);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_overflow_extents(const uint64_t* overflow, int64_t K, int64_t* ext, void* stream) {
    DME_REQUIRE(overflow && ext && K >= 1, "bad arguments");
    hipLaunchKernelGGL(overflow_extents_kernel, dim3(256), dim3(SHARD_BLOCK), 0, (hipStream_t)stream,
                       (const u64*)overflow, (long long)K, (long long*)ext);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_overflow_append(uint64_t* dst, const uint64_t* src, uint64_t* status, void* stream) {
    DME_REQUIRE(dst && src && dst != src, "bad arguments");
    hipLaunchKernelGGL(overflow_append_kernel, dim3(256), dim3(SHARD_BLOCK), 0, (hipStream_t)stream,
                       (u64*)dst, (const u64*)src, (u64*)status);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_ragged_plan(const int64_t* ext, int64_t K, int64_t* off, int64_t* slotoff,
                               void* stream) {
    DME_REQUIRE(ext && off && slotoff && K >= 1, "bad arguments");
    hipLaunchKernelGGL(ragged_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       (const long long*)ext, (long long)K, 256LL, (long long*)off, (long long*)slotoff);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int64_t dme_shard_message_words32(int64_t nh, int64_t nv, int64_t K, int64_t world) {
    if (nh < 0 || nv < 0 || K < 1 || world < 1) return -1;
    return nh + nv + 2 * world * (8 + K);
}

extern "C" int dme_shard_pack32(const uint64_t* hist, int64_t nh, const uint64_t* histV, int64_t nv,
                                const uint64_t* status, const double* init_acc, int64_t K,
                                int64_t rank, int64_t world, uint32_t* msg, void* stream) {
    DME_REQUIRE(nh >= 0 && nv >= 0 && K >= 1, "bad sizes nh=%lld nv=%lld K=%lld", (long long)nh,
                (long long)nv, (long long)K);
    DME_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %lld of %lld", (long long)rank,
                (long long)world);
    DME_REQUIRE((nh == 0 || hist) && (nv == 0 || histV) && status && init_acc && msg, "null pointer");
    const long long total = nh + nv + 2 * world * (8 + K);
    long long grid = (total + SHARD_BLOCK * 4 - 1) / (SHARD_BLOCK * 4);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(shard_pack32_kernel, dim3((unsigned)grid), dim3(SHARD_BLOCK), 0,
                       (hipStream_t)stream, (const u64*)hist, (long long)nh, (const u64*)histV,
                       (long long)nv, (const u64*)status, init_acc, (long long)K, (long long)rank,
                       (long long)world, (u32*)msg);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_shard_unpack32(const uint32_t* msg, int64_t nh, int64_t nv, int64_t K,
                                  int64_t world, uint64_t* hist64, uint64_t* histV64,
                                  uint64_t* status, double* init_acc, void* stream) {
    DME_REQUIRE(nh >= 0 && nv >= 0 && K >= 1 && world >= 1, "bad sizes");
    DME_REQUIRE(msg && (nh == 0 || hist64) && (nv == 0 || histV64) && status && init_acc, "null pointer");
    long long n = nh + nv > K ? nh + nv : K;
    if (n < 8) n = 8;
    long long grid = (n + SHARD_BLOCK * 4 - 1) / (SHARD_BLOCK * 4);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(shard_unpack32_kernel, dim3((unsigned)grid), dim3(SHARD_BLOCK), 0,
                       (hipStream_t)stream, (const u32*)msg, (long long)nh, (long long)nv,
                       (long long)K, (long long)world, (u64*)hist64, (u64*)histV64, (u64*)status,
                       init_acc);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int64_t dme_shard_message_words(int64_t nh, int64_t nv, int64_t K, int64_t world) {
    if (nh < 0 || nv < 0 || K < 1 || world < 1) return -1;
    return nh + nv + 8 + world * (2 + K);
}

extern "C" int dme_shard_pack(const uint64_t* hist, int64_t nh, const uint64_t* histV, int64_t nv,
                              const uint64_t* status, const double* init_acc, int64_t K,
                              int64_t rank, int64_t world, uint64_t* msg, void* stream) {
    DME_REQUIRE(nh >= 0 && nv >= 0 && K >= 1, "bad sizes nh=%lld nv=%lld K=%lld", (long long)nh,
                (long long)nv, (long long)K);
    DME_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %lld of %lld", (long long)rank,
                (long long)world);
    DME_REQUIRE((nh == 0 || hist) && (nv == 0 || histV) && status && init_acc && msg, "null pointer");
    const long long total = nh + nv + 8 + world * (2 + K);
    long long grid = (total + SHARD_BLOCK * 4 - 1) / (SHARD_BLOCK * 4);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned)grid), dim3(SHARD_BLOCK), 0,
                       (hipStream_t)stream, (const u64*)hist, (long long)nh, (const u64*)histV,
                       (long long)nv, (const u64*)status, init_acc, (long long)K, (long long)rank,
                       (long long)world, (u64*)msg);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_shard_unpack(uint64_t* msg, int64_t nh, int64_t nv, int64_t K, int64_t world,
                                double* init_acc, void* stream) {
    DME_REQUIRE(nh >= 0 && nv >= 0 && K >= 1 && world >= 1, "bad sizes");
    DME_REQUIRE(msg && init_acc, "null pointer");
    const long long n = K > 2 ? K : 2;
    hipLaunchKernelGGL(shard_unpack_kernel, dim3((unsigned)((n + SHARD_BLOCK - 1) / SHARD_BLOCK)),
                       dim3(SHARD_BLOCK), 0, (hipStream_t)stream, (u64*)msg, (long long)nh,
                       (long long)nv, (long long)K, (long long)world, init_acc);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
