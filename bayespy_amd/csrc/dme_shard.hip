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

}  // namespace

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
