// dme_bounds.hip — largest cell and largest row total of a dense count matrix.
//
// The reference grows U[k] / V as rows arrive (DE/dme.py:118, :122 append when the count or the row
// total exceeds the current length); the statistics pass here writes into histograms whose extents
// (nbins = largest count + 1, tbins = largest row total + 1) must be known beforehand.  A caller
// who does not know them measures them with this pass: one streaming read of X (wave per row,
// 16-byte loads when the layout allows), negative cells ignored (the statistics pass reports them).
#include "dme_common.h"

namespace {

typedef u32 bx4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void bounds_cell(T x, u64& s, u32& mc) {
    if constexpr (sizeof(T) == 2) {
        const u32 v = (u32)x;
        s += v;
        mc = v > mc ? v : mc;
    } else {
        const int v = (int)x;
        if (v > 0) {
            s += (u64)v;
            mc = (u32)v > mc ? (u32)v : mc;
        }
    }
}

// VEC: rows are read as 16-byte vectors (K, ld multiples of 16 / sizeof(T), X 16-byte aligned)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bounds_kernel(const T* X, long long N, long long K,
                                                     long long ld, u64* out) {
    constexpr int E = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    u32 mc = 0;
    u64 mt = 0;
    for (long long i = wave0; i < N; i += nwaves) {
        const T* row = X + i * ld;
        u64 s = 0;
        if constexpr (VEC) {
            const long long C = K / E;
            for (long long c = lane; c < C; c += 64) {
                const bx4 v = __builtin_nontemporal_load((const bx4*)(row + c * E));
                const u32 d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (sizeof(T) == 2) {
                        bounds_cell<unsigned short>((unsigned short)(d[q] & 0xffffu), s, mc);
                        bounds_cell<unsigned short>((unsigned short)(d[q] >> 16), s, mc);
                    } else {
                        bounds_cell<int>((int)d[q], s, mc);
                    }
                }
            }
        } else {
            for (long long k = lane; k < K; k += 64) bounds_cell<T>(row[k], s, mc);
        }
        const u64 total = dme_wave_sum_u64(s);
        mt = total > mt ? total : mt;
    }
    const u64 m = dme_wave_max_u64((u64)mc);
    if (lane == 0) {
        if (m) atomicMax(&out[0], m);
        if (mt) atomicMax(&out[1], mt);
    }
}

template <typename T>
int bounds_dense(const T* X, long long N, long long K, long long ld, u64* out, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K, "bad shape N=%lld K=%lld ld=%lld", N, K, ld);
    DME_REQUIRE(out != nullptr, "null output pointer");
    if (N == 0) return DME_OK;
    DME_REQUIRE(X != nullptr, "X is null");
    DME_REQUIRE(((uintptr_t)X % sizeof(T)) == 0, "X is not element-aligned");
    constexpr int E = 16 / (int)sizeof(T);
    const bool vec = K % E == 0 && ld % E == 0 && ((uintptr_t)X % 16) == 0;
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    long long waves = N < (long long)cus * 32 ? N : (long long)cus * 32;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL((bounds_kernel<T, true>), dim3(grid), dim3(256), 0, s, X, N, K, ld, out);
    else hipLaunchKernelGGL((bounds_kernel<T, false>), dim3(grid), dim3(256), 0, s, X, N, K, ld, out);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

// CSR rows: a wave takes 64 consecutive rows, one per lane when they are short (the bag-of-words
// case: tens of entries), and walks long rows together.
__global__ __launch_bounds__(256) void bounds_csr_kernel(const long long* rowptr, const int* vals,
                                                         long long N, long long nnz, u64* out) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    u32 mc = 0;
    u64 mt = 0;
    for (long long r0 = wave0 * 64; r0 < N; r0 += nwaves * 64) {
        const long long r = r0 + lane;
        long long p0 = 0, p1 = 0;
        if (r < N) {
            p0 = rowptr[r];
            p1 = rowptr[r + 1];
            if (p0 < 0) p0 = 0;
            if (p1 > nnz) p1 = nnz;  // an inconsistent rowptr is reported by the statistics pass
        }
        u64 s = 0;
        for (long long p = p0; p < p1; ++p) {
            const int v = vals[p];
            if (v > 0) {
                s += (u64)v;
                mc = (u32)v > mc ? (u32)v : mc;
            }
        }
        mt = s > mt ? s : mt;
    }
    const u64 m = dme_wave_max_u64((u64)mc);
    const u64 t = dme_wave_max_u64(mt);
    if (lane == 0) {
        if (m) atomicMax(&out[0], m);
        if (t) atomicMax(&out[1], t);
    }
}

}  // namespace

extern "C" int dme_bounds_csr_i32(const int64_t* rowptr, const int32_t* vals, int64_t N,
                                  int64_t nnz, uint64_t* out, void* stream) {
    DME_REQUIRE(N >= 0 && nnz >= 0, "bad shape N=%lld nnz=%lld", (long long)N, (long long)nnz);
    DME_REQUIRE(out != nullptr, "null output pointer");
    if (N == 0 || nnz == 0) return DME_OK;
    DME_REQUIRE(rowptr && vals, "null CSR pointer");
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    long long waves = (N + 63) / 64;
    if (waves > (long long)cus * 64) waves = (long long)cus * 64;
    hipLaunchKernelGGL(bounds_csr_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const long long*)rowptr, vals, (long long)N,
                       (long long)nnz, (u64*)out);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_bounds_dense_u16(const uint16_t* X, int64_t N, int64_t K, int64_t ld,
                                    uint64_t* out, void* stream) {
    return bounds_dense<unsigned short>(X, N, K, ld, (u64*)out, stream);
}

extern "C" int dme_bounds_dense_i32(const int32_t* X, int64_t N, int64_t K, int64_t ld,
                                    uint64_t* out, void* stream) {
    return bounds_dense<int>(X, N, K, ld, (u64*)out, stream);
}
