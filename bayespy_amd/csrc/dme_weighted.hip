// dme_weighted.hip — statistics of rows that carry a weight.
//
// CompressedRowData.appendRow(row, weight) (DirichletEstimation/dirichletMultinomialEstimation.py
// :113-123) adds `weight`, not 1, to U[k][j] (j < row[k]) and V[j] (j < sum(row)); the reference's
// Dirichlet-mixture EM calls it with fractional responsibilities
// (ConjugatePriorTools/dirichletMixtureEstimation.py:129-133).  In histogram form:
//     histW[k][v] += weight for x_ik == v,   histVW[t] += weight for sum_k x_ik == t      (fp64)
// and U/V are again reverse suffix sums.  len(U[k]) grows with every appended row whatever its
// weight, so the largest count per column / largest total are tracked as integers beside the sums.
// First version: one wave per row, fp64 global atomics (no LDS privatisation) — the callers of
// this path hold thousands of rows, not billions.  fp64 sums here depend on arrival order in the
// last bits (the reference adds row by row).
#include "dme_common.h"

namespace {

__global__ __launch_bounds__(256) void stats_weighted_kernel(
    const int* X, const double* wts, long long N, long long K, long long ld, double* histW,
    long long nbins, double* histVW, long long tbins, long long* maxcol, long long* maxtot,
    u64* status) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    u64 bad_cells = 0, bad_rows = 0, neg = 0, rows = 0;
    for (long long i = wave0; i < N; i += nwaves) {
        const int* row = X + i * ld;
        const double w = wts[i];
        u64 s = 0;
        for (long long k = lane; k < K; k += 64) {
            const int v = row[k];
            s += (u64)(v > 0 ? v : 0);
        }
        const u64 total = dme_wave_sum_u64(s);
        rows++;
        if (lane == 0 && total > 0) {
            if ((long long)total < tbins) {
                atomicAdd(&histVW[total], w);
                // the maxima settle after a few rows: look before paying for an atomic
                if ((u64)__builtin_nontemporal_load(maxtot) < total) atomicMax((u64*)maxtot, total);
            } else {
                bad_rows++;
            }
        }
        for (long long k = lane; k < K; k += 64) {
            const int v = row[k];
            if (v == 0) continue;
            if (v < 0) { neg++; continue; }
            if ((long long)v < nbins) {
                atomicAdd(&histW[k * nbins + v], w);
                if (__builtin_nontemporal_load(&maxcol[k]) < (long long)v) atomicMax((u64*)&maxcol[k], (u64)v);
            } else {
                bad_cells++;
            }
        }
    }
    bad_cells = dme_wave_sum_u64(bad_cells);
    bad_rows = dme_wave_sum_u64(bad_rows);
    neg = dme_wave_sum_u64(neg);
    if (lane == 0) {
        if (rows) atomicAdd(&status[0], rows);
        if (bad_cells) atomicAdd(&status[1], bad_cells);
        if (bad_rows) atomicAdd(&status[2], bad_rows);
        if (neg) atomicAdd(&status[3], neg);
    }
}

// Uf[k][n] += sum_{v > n} histW[k][v]; lenU[k] = max(lenU[k], maxcol[k]).  One thread per column
// walks the bins from the top (the weighted statistics are small); column K is V.
__global__ __launch_bounds__(256) void suffix_scan_weighted_kernel(
    const double* histW, long long K, long long nbins, const double* histVW, long long tbins,
    const long long* maxcol, const long long* maxtot, double* Uf, long long ldU, long long* lenU,
    double* Vf, long long capV, long long* lenV) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > K) return;
    const bool isV = k == K;
    const double* src = isV ? histVW : histW + k * nbins;
    const long long top = isV ? maxtot[0] : maxcol[k];
    double* out = isV ? Vf : Uf + k * ldU;
    const long long cap = isV ? capV : ldU;
    double run = 0.0;
    for (long long v = top; v >= 1; --v) {
        run = __dadd_rn(run, src[v]);
        if (v - 1 < cap) out[v - 1] = __dadd_rn(out[v - 1], run);
    }
    long long* len = isV ? lenV : &lenU[k];
    if (top > *len) *len = top;
}

}  // namespace

extern "C" int dme_stats_dense_weighted_i32(const int32_t* X, const double* weights, int64_t N,
                                            int64_t K, int64_t ld, double* histW, int64_t nbins,
                                            double* histVW, int64_t tbins, int64_t* maxcol,
                                            int64_t* maxtot, uint64_t* status, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K, "bad shape N=%lld K=%lld ld=%lld", (long long)N,
                (long long)K, (long long)ld);
    DME_REQUIRE(nbins >= 1 && tbins >= 1, "nbins/tbins must be >= 1");
    DME_REQUIRE(histW && histVW && maxcol && maxtot && status, "null output pointer");
    if (N == 0) return DME_OK;
    DME_REQUIRE(X && weights, "null input pointer");
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    long long waves = N < (long long)cus * 16 ? N : (long long)cus * 16;
    hipLaunchKernelGGL(stats_weighted_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, X, weights, (long long)N, (long long)K, (long long)ld,
                       histW, (long long)nbins, histVW, (long long)tbins, (long long*)maxcol,
                       (long long*)maxtot, (u64*)status);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_suffix_scan_weighted(const double* histW, int64_t K, int64_t nbins,
                                        const double* histVW, int64_t tbins, const int64_t* maxcol,
                                        const int64_t* maxtot, double* Uf, int64_t ldU,
                                        int64_t* lenU, double* Vf, int64_t capV, int64_t* lenV,
                                        void* stream) {
    DME_REQUIRE(K >= 1 && nbins >= 1 && tbins >= 1, "bad shape");
    DME_REQUIRE(ldU >= nbins - 1 && capV >= tbins - 1, "U/V too short for the histograms");
    DME_REQUIRE(histW && histVW && maxcol && maxtot && Uf && lenU && Vf && lenV, "null pointer");
    hipLaunchKernelGGL(suffix_scan_weighted_kernel, dim3((unsigned)((K + 1 + 255) / 256)), dim3(256),
                       0, (hipStream_t)stream, histW, (long long)K, (long long)nbins, histVW,
                       (long long)tbins, (const long long*)maxcol, (const long long*)maxtot, Uf,
                       (long long)ldU, (long long*)lenU, Vf, (long long)capV, (long long*)lenV);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
