// dme_dense2csr.hip — dense count rows wider than the dense statistics kernels take (uint16: more
// than 32 768 columns, int32: more than 16 384) are compressed to CSR on the device and go through
// dme_stats_csr_i32.  No counterpart in the reference, whose appendRow takes a row of any length
// (DE/dme.py:113-123): this is what keeps the drop-in's appendRow / appendMatrix working for
// vocabularies of BASELINE configs[3]'s size when the caller holds dense rows.
//
//   dme_dense_row_nnz_*   counts[i] = #{k : x_ik != 0}                      (one wave per row)
//   (dme_shard_offsets    rowptr = prefix sums of counts: dme_shard.hip's single-workgroup scan)
//   dme_dense_to_csr_*    colidx / vals of the non-zero cells, row by row, columns ascending
// Negative cells are kept (they are not zero): the CSR pass reports them as the dense pass would.
#include "dme_common.h"

namespace {

constexpr int D2C_BLOCK = 256;
constexpr int D2C_ROWS = D2C_BLOCK / 64;

template <typename T>
__global__ __launch_bounds__(D2C_BLOCK) void dense_row_nnz_kernel(const T* X, long long N, long long K,
                                                                  long long ld, long long* counts) {
    const int lane = threadIdx.x & 63;
    for (long long i = (long long)blockIdx.x * D2C_ROWS + (threadIdx.x >> 6); i < N;
         i += (long long)gridDim.x * D2C_ROWS) {
        const T* row = X + i * ld;
        u32 n = 0;
        for (long long k = lane; k < K; k += 64) n += row[k] != 0 ? 1u : 0u;
        n = dme_wave_sum_u32(n);
        if (lane == 0) counts[i] = (long long)n;
    }
}

template <typename T>
__global__ __launch_bounds__(D2C_BLOCK) void dense_to_csr_kernel(const T* X, long long N, long long K,
                                                                 long long ld, const long long* rowptr,
                                                                 int* colidx, int* vals) {
    const int lane = threadIdx.x & 63;
    const u64 below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));  // lanes before this one
    for (long long i = (long long)blockIdx.x * D2C_ROWS + (threadIdx.x >> 6); i < N;
         i += (long long)gridDim.x * D2C_ROWS) {
        const T* row = X + i * ld;
        long long out = rowptr[i];
        for (long long k0 = 0; k0 < K; k0 += 64) {  // wave-uniform trip count
            const long long k = k0 + lane;
            const long long v = k < K ? (long long)row[k] : 0;
            const u64 m = __ballot(v != 0);
            if (v != 0) {
                const long long o = out + __popcll(m & below);
                colidx[o] = (int)k;
                vals[o] = (int)v;
            }
            out += __popcll(m);
        }
    }
}

long long d2c_grid(long long N) {
    long long g = (N + D2C_ROWS - 1) / D2C_ROWS;
    if (g > 16384) g = 16384;
    return g < 1 ? 1 : g;
}

template <typename T>
int row_nnz(const T* X, long long N, long long K, long long ld, long long* counts, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K && K < (1LL << 31), "bad shape N=%lld K=%lld ld=%lld", N, K, ld);
    if (N == 0) return DME_OK;
    DME_REQUIRE(X && counts, "null pointer");
    hipLaunchKernelGGL(dense_row_nnz_kernel<T>, dim3((unsigned)d2c_grid(N)), dim3(D2C_BLOCK), 0,
                       (hipStream_t)stream, X, N, K, ld, counts);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

template <typename T>
int to_csr(const T* X, long long N, long long K, long long ld, const long long* rowptr, int* colidx,
           int* vals, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K && K < (1LL << 31), "bad shape N=%lld K=%lld ld=%lld", N, K, ld);
    if (N == 0) return DME_OK;
    DME_REQUIRE(X && rowptr && colidx && vals, "null pointer");
    hipLaunchKernelGGL(dense_to_csr_kernel<T>, dim3((unsigned)d2c_grid(N)), dim3(D2C_BLOCK), 0,
                       (hipStream_t)stream, X, N, K, ld, rowptr, colidx, vals);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

}  // namespace

extern "C" int dme_dense_row_nnz_u16(const uint16_t* X, int64_t N, int64_t K, int64_t ld,
                                     int64_t* counts, void* stream) {
    return row_nnz<unsigned short>(X, N, K, ld, (long long*)counts, stream);
}
extern "C" int dme_dense_row_nnz_i32(const int32_t* X, int64_t N, int64_t K, int64_t ld,
                                     int64_t* counts, void* stream) {
    return row_nnz<int>(X, N, K, ld, (long long*)counts, stream);
}
extern "C" int dme_dense_to_csr_u16(const uint16_t* X, int64_t N, int64_t K, int64_t ld,
                                    const int64_t* rowptr, int32_t* colidx, int32_t* vals,
                                    void* stream) {
    return to_csr<unsigned short>(X, N, K, ld, (const long long*)rowptr, colidx, vals, stream);
}
extern "C" int dme_dense_to_csr_i32(const int32_t* X, int64_t N, int64_t K, int64_t ld,
                                    const int64_t* rowptr, int32_t* colidx, int32_t* vals,
                                    void* stream) {
    return to_csr<int>(X, N, K, ld, (const long long*)rowptr, colidx, vals, stream);
}
