// dme_common.h — internal helpers shared by the HIP translation units of libdme_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dme_hip.h"

// Sets the thread-local error text and returns `code` (defined in dme_capi.hip).
int dme_set_error(int code, const char* fmt, ...);

#define DME_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return dme_set_error(DME_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define DME_REQUIRE(cond, ...)                                              \
    do {                                                                    \
        if (!(cond)) return dme_set_error(DME_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

// ---------------------------------------------------------------------------------------------
// Overflow store (include/dme_hip.h, dme_overflow_*): the cells of a statistics pass whose count does
// not fit the dense histogram, hist[k][v] with v >= nbins, as records.  One outlier count then costs 16
// bytes instead of widening every column (the reference's U[k] are lists of their own lengths,
// DE/dme.py:118).  Layout: u64 head[2] = {used, capacity}, then `capacity` records.  `used` counts every
// attempt, also those beyond the capacity (which the caller reports in status[1], "not counted").
// ---------------------------------------------------------------------------------------------
struct DmeOvfRec {
    u32 k, v;  // column, count value
    u64 n;     // how many rows (a statistics pass appends 1; merged stores hold sums)
};
static_assert(sizeof(DmeOvfRec) == 16, "record layout is part of the ABI");

__device__ __forceinline__ bool dme_overflow_put(u64* ovf, u64 k, u64 v, u64 n) {
    if (!ovf || v > 0xFFFFFFFFULL) return false;
    const u64 i = atomicAdd(&ovf[0], 1ULL);
    if (i >= ovf[1]) return false;
    DmeOvfRec* r = (DmeOvfRec*)(ovf + 2) + i;
    r->k = (u32)k;
    r->v = (u32)v;
    r->n = n;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Wavefront (64 lanes) reductions on DPP: xor-1 / xor-2 inside a quad, then the two mirrors
// leave every lane of a 16-lane row holding the row sum; the four row sums are combined through
// scalar registers (v_readlane), which also makes the result wave-uniform.
// ---------------------------------------------------------------------------------------------
#define DME_DPP_QUAD_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define DME_DPP_QUAD_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define DME_DPP_HALF_MIRROR 0x141
#define DME_DPP_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ u32 dme_dpp_u32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ double dme_dpp_f64(double v) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, false);
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, false);
    return b.d;
}

__device__ __forceinline__ u32 dme_wave_sum_u32(u32 v) {
    v += dme_dpp_u32<DME_DPP_QUAD_XOR1>(v);
    v += dme_dpp_u32<DME_DPP_QUAD_XOR2>(v);
    v += dme_dpp_u32<DME_DPP_HALF_MIRROR>(v);
    v += dme_dpp_u32<DME_DPP_MIRROR>(v);
    return (u32)__builtin_amdgcn_readlane((int)v, 0) + (u32)__builtin_amdgcn_readlane((int)v, 16) +
           (u32)__builtin_amdgcn_readlane((int)v, 32) + (u32)__builtin_amdgcn_readlane((int)v, 48);
}

// Inclusive prefix sum over the 64 lanes: four row_shr steps inside the rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 carry the rows' sums over (lanes without a source add 0).  VALU only.
__device__ __forceinline__ u32 dme_wave_incl_scan_u32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

__device__ __forceinline__ u32 dme_wave_max_u32(u32 v) {
    v = max(v, dme_dpp_u32<DME_DPP_QUAD_XOR1>(v));
    v = max(v, dme_dpp_u32<DME_DPP_QUAD_XOR2>(v));
    v = max(v, dme_dpp_u32<DME_DPP_HALF_MIRROR>(v));
    v = max(v, dme_dpp_u32<DME_DPP_MIRROR>(v));
    u32 a = max((u32)__builtin_amdgcn_readlane((int)v, 0), (u32)__builtin_amdgcn_readlane((int)v, 16));
    u32 b = max((u32)__builtin_amdgcn_readlane((int)v, 32), (u32)__builtin_amdgcn_readlane((int)v, 48));
    return max(a, b);
}

__device__ __forceinline__ u64 dme_wave_sum_u64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ u64 dme_wave_max_u64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double dme_readlane_f64(double v, int lane) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    b.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return b.d;
}

// Error-free transformation: s + e == a + b exactly (Knuth two-sum).
__device__ __forceinline__ void dme_two_sum(double a, double b, double& s, double& e) {
    s = __dadd_rn(a, b);
    double bb = __dsub_rn(s, a);
    e = __dadd_rn(__dsub_rn(a, __dsub_rn(s, bb)), __dsub_rn(b, bb));
}

// Compensated accumulator: value = hi + lo.
struct DmeAcc {
    double hi, lo;
    __device__ __forceinline__ void add(double x) {
        double s, e;
        dme_two_sum(hi, x, s, e);
        hi = s;
        lo = __dadd_rn(lo, e);
    }
    __device__ __forceinline__ void add(const DmeAcc& o) {
        double s, e;
        dme_two_sum(hi, o.hi, s, e);
        hi = s;
        lo = __dadd_rn(lo, __dadd_rn(o.lo, e));
    }
    __device__ __forceinline__ double value() const { return __dadd_rn(hi, lo); }
};

template <int CTRL>
__device__ __forceinline__ DmeAcc dme_dpp_acc(const DmeAcc& a) {
    DmeAcc r;
    r.hi = dme_dpp_f64<CTRL>(a.hi);
    r.lo = dme_dpp_f64<CTRL>(a.lo);
    return r;
}

// Full-wave compensated sum; the result (wave-uniform) is returned on every lane.
__device__ __forceinline__ DmeAcc dme_wave_sum_acc(DmeAcc v) {
    v.add(dme_dpp_acc<DME_DPP_QUAD_XOR1>(v));
    v.add(dme_dpp_acc<DME_DPP_QUAD_XOR2>(v));
    v.add(dme_dpp_acc<DME_DPP_HALF_MIRROR>(v));
    v.add(dme_dpp_acc<DME_DPP_MIRROR>(v));
    DmeAcc r;
    r.hi = dme_readlane_f64(v.hi, 0);
    r.lo = dme_readlane_f64(v.lo, 0);
#pragma unroll
    for (int row = 16; row < 64; row += 16) {
        DmeAcc o;
        o.hi = dme_readlane_f64(v.hi, row);
        o.lo = dme_readlane_f64(v.lo, row);
        r.add(o);
    }
    return r;
}

__device__ __forceinline__ double dme_wave_sum_f64(double x) {
    x = __dadd_rn(x, dme_dpp_f64<DME_DPP_QUAD_XOR1>(x));
    x = __dadd_rn(x, dme_dpp_f64<DME_DPP_QUAD_XOR2>(x));
    x = __dadd_rn(x, dme_dpp_f64<DME_DPP_HALF_MIRROR>(x));
    x = __dadd_rn(x, dme_dpp_f64<DME_DPP_MIRROR>(x));
    double r = dme_readlane_f64(x, 0);
    r = __dadd_rn(r, dme_readlane_f64(x, 16));
    r = __dadd_rn(r, dme_readlane_f64(x, 32));
    r = __dadd_rn(r, dme_readlane_f64(x, 48));
    return r;
}

// DE/fdp.py:104-105 `priorSum = sum(priors) + 0.01`: the mean accumulators summed in a fixed order by
// a BLOCK-thread workgroup (strided per thread, DPP inside a wave, waves in order).  Shared by
// dme_cli_init_alphas and by the solver's set-up when it starts from the accumulators, so that both
// give the same bits.  Every thread gets the result; wsum is LDS, BLOCK / 64 doubles.
template <int BLOCK>
__device__ __forceinline__ double dme_cli_prior_sum(const double* acc, long long K, double* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (long long k = tid; k < K; k += BLOCK) s = __dadd_rn(s, acc[k]);
    s = dme_wave_sum_f64(s);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) tot = __dadd_rn(tot, wsum[w]);
    __syncthreads();
    return __dadd_rn(tot, 0.01);
}
