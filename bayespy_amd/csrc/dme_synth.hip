// dme_synth.hip — synthetic Dirichlet-multinomial count rows, generated on the device.
//
// Benchmark / test input only (not on the measured path).  The reference's generators are
// host-side samplers (DirichletEstimation/samplingTools.py:41-67: gamma draws for the Dirichlet,
// then one categorical draw per count); at N = 1e7 x K = 1000 that is far too slow, so rows are
// drawn here by the equivalent Polya-urn scheme, which needs no gamma variates:
//   draw j (j earlier draws) copies a uniformly chosen earlier draw with probability j / (A + j),
//   otherwise it is a fresh draw from alpha / A  (binary search in the prefix sums of alpha).
// The counts of n such draws are exactly DirichletMultinomial(n, alpha).
// One thread per row; the draw history lives in a transposed scratch array so that the sequential
// writes of a wave are coalesced.  RNG: splitmix64 keyed on (seed, absolute row index), so any row
// range reproduces the same rows whatever the launch shape.
#include "dme_common.h"

namespace {

__device__ __forceinline__ u64 splitmix(u64& st) {
    u64 z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <typename T>
__global__ __launch_bounds__(64) void synth_kernel(T* X, long long N, long long K, long long ld,
                                                   const int* totals, long long max_total,
                                                   const double* cdf, u64 seed,
                                                   long long row_offset, u32* hist,
                                                   long long nthreads) {
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    if (t >= nthreads) return;
    const double A = cdf[K - 1];
    for (long long i = t; i < N; i += nthreads) {
        long long n = totals[i];
        if (n > max_total) n = max_total;
        u64 st = seed ^ ((u64)(row_offset + i) * 0xD1342543DE82EF95ULL);
        splitmix(st);
        T* row = X + i * ld;
        for (long long j = 0; j < n; ++j) {
            const double u1 = (double)(splitmix(st) >> 11) * 0x1.0p-53;
            const u64 r2 = splitmix(st);
            u32 cat;
            if (u1 * (A + (double)j) < (double)j) {
                // copy an earlier draw
                const long long pick = (long long)(((r2 >> 32) * (u64)j) >> 32);
                cat = hist[pick * nthreads + t];
            } else {
                const double target = (double)(r2 >> 11) * 0x1.0p-53 * A;
                long long lo = 0, hi = K - 1;
                while (lo < hi) {
                    const long long mid = (lo + hi) >> 1;
                    if (cdf[mid] > target) hi = mid;
                    else lo = mid + 1;
                }
                cat = (u32)lo;
            }
            hist[j * nthreads + t] = cat;
            row[cat] = (T)(row[cat] + 1);
        }
    }
}

long long synth_threads(long long max_total) {
    if (max_total < 1) max_total = 1;
    long long t = (1LL << 30) / (max_total * 4);
    if (t > 262144) t = 262144;
    if (t < 1024) t = 1024;
    return t & ~63LL;
}

}  // namespace

extern "C" int64_t dme_synth_scratch_bytes(int64_t max_total) {
    if (max_total < 1) max_total = 1;
    return synth_threads(max_total) * max_total * 4 + 256;
}

extern "C" int dme_synth_dirmult(void* X, int elem_bytes, int64_t N, int64_t K, int64_t ld,
                                 const int32_t* totals, int64_t max_total, const double* cdf,
                                 uint64_t seed, int64_t row_offset, void* scratch,
                                 int64_t scratch_bytes, void* stream) {
    DME_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "elem_bytes must be 2 or 4");
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K && max_total >= 0, "bad shape");
    DME_REQUIRE(elem_bytes == 4 || max_total <= 65535, "uint16 rows need max_total <= 65535");
    if (N == 0) return DME_OK;
    DME_REQUIRE(X && totals && cdf && scratch, "null pointer");
    long long nthreads = synth_threads(max_total);
    if (scratch_bytes < nthreads * (max_total > 0 ? max_total : 1) * 4)
        return dme_set_error(DME_ERR_WORKSPACE, "synth scratch too small");
    if (nthreads > ((N + 63) & ~63LL)) nthreads = (N + 63) & ~63LL;
    hipStream_t s = (hipStream_t)stream;
    DME_HIP_CHECK(hipMemsetAsync(X, 0, (size_t)N * ld * elem_bytes, s));
    const unsigned grid = (unsigned)(nthreads / 64);
    if (elem_bytes == 2)
        hipLaunchKernelGGL(synth_kernel<unsigned short>, dim3(grid), dim3(64), 0, s,
                           (unsigned short*)X, (long long)N, (long long)K, (long long)ld, totals,
                           (long long)max_total, cdf, (u64)seed, (long long)row_offset,
                           (u32*)scratch, nthreads);
    else
        hipLaunchKernelGGL(synth_kernel<int>, dim3(grid), dim3(64), 0, s, (int*)X, (long long)N,
                           (long long)K, (long long)ld, totals, (long long)max_total, cdf,
                           (u64)seed, (long long)row_offset, (u32*)scratch, nthreads);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
