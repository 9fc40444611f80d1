// dme_scan.hip — reverse suffix scan: value histograms -> the reference's compressed statistics.
//
// CompressedRowData.appendRow (DirichletEstimation/dirichletMultinomialEstimation.py:116-123) adds
// 1 to U[k][j] for every j < x_ik and to V[j] for every j < sum_k x_ik, i.e.
//     U[k][n] = #{i : x_ik > n} = sum_{v > n} hist[k][v],      len(U[k]) = max_i x_ik
//     V[n]    = #{i : n_i  > n} = sum_{t > n} histV[t],        len(V)    = max_i n_i
// so U/V are the reverse inclusive scans of the value histograms with bin 0 dropped.  Integer work,
// exact.  One 256-thread workgroup per column (column K is V); tiles are walked from the top bin
// down with a running carry.  Also holds the CLI's init normalisation (DE/fdp.py:104-108).
#include "dme_common.h"

namespace {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_IT = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_IT;

// RAGGED (off != NULL): `hist` holds the bins 1 .. off[k+1] - off[k] of every column laid end to end
// (dme_shard_compact) and U / Uf take the same places, U[k][n] at off[k] + n: a column costs its own
// length.  slotcol (with slotoff, dme_ragged_plan) receives the column of every 256-term slot of the
// solver's ragged layout.  Columns are dealt to the workgroups round robin (K may exceed the grid).
__global__ __launch_bounds__(SCAN_BLOCK) void suffix_scan_kernel(
    const u64* hist, long long K, long long nbins, const u64* histV, long long tbins, long long* U,
    double* Uf, long long ldU, long long* lenU, long long* V, double* Vf, long long* lenV,
    const long long* off, const long long* slotoff, int* slotcol) {
    __shared__ u64 wave_tot[SCAN_BLOCK / 64];
    __shared__ u64 wave_max[SCAN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (long long k = blockIdx.x; k <= K; k += gridDim.x) {
    const bool isV = k >= K;
    const long long o = (off && !isV) ? off[k] : 0;
    const long long ext = (off && !isV) ? off[k + 1] - o : 0;
    // src[v] = the column's bin v
    const u64* src = isV ? histV : (off ? hist + o - 1 : hist + k * nbins);
    const long long nb = isV ? tbins : (off ? ext + 1 : nbins);
    long long* outI = isV ? V : (off ? U + o : U + k * ldU);
    double* outF = isV ? Vf : (Uf ? (off ? Uf + o : Uf + k * ldU) : nullptr);
    const long long cap = isV ? tbins - 1 : (off ? ext : ldU);  // entries available in the output row
    if (off && !isV && slotcol) {
        const long long s0 = slotoff[k], s1 = slotoff[k + 1];
        for (long long i = s0 + tid; i < s1; i += SCAN_BLOCK) slotcol[i] = (int)k;
    }

    u64 carry = 0;   // sum of all bins above the current tile
    u64 maxnz = 0;   // largest non-empty bin seen by this thread
    for (long long hi = nb; hi > 1; hi -= SCAN_TILE) {
        // this thread: bins v = hi-1 - (tid*IT + i), i = 0..IT-1 (descending), v >= 1 only
        u64 x[SCAN_IT], incl[SCAN_IT];
        u64 run = 0;
#pragma unroll
        for (int i = 0; i < SCAN_IT; ++i) {
            const long long v = hi - 1 - ((long long)tid * SCAN_IT + i);
            x[i] = (v >= 1) ? src[v] : 0ULL;
            if (x[i] && (u64)v > maxnz) maxnz = (u64)v;
            run += x[i];
            incl[i] = run;
        }
        // block exclusive scan of `run` in thread order (= descending bins)
        u64 sc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u64 up = __shfl_up(sc, d);
            if (lane >= d) sc += up;
        }
        if (lane == 63) wave_tot[wave] = sc;
        __syncthreads();
        u64 before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
            const u64 t = wave_tot[w];
            if (w < wave) before += t;
            total += t;
        }
        const u64 excl = carry + before + (sc - run);
#pragma unroll
        for (int i = 0; i < SCAN_IT; ++i) {
            const long long v = hi - 1 - ((long long)tid * SCAN_IT + i);
            if (v >= 1 && v - 1 < cap) {
                const u64 sfx = excl + incl[i];  // sum_{v' >= v} src[v']  ==  U[v-1]
                outI[v - 1] = (long long)sfx;
                if (outF) outF[v - 1] = (double)sfx;
            }
        }
        carry += total;
        __syncthreads();
    }
    // padding beyond nbins-1 (ldU may be larger than nbins-1)
    for (long long n = (nb >= 1 ? nb - 1 : 0) + tid; n < cap; n += SCAN_BLOCK) {
        outI[n] = 0;
        if (outF) outF[n] = 0.0;
    }
    u64 m = dme_wave_max_u64(maxnz);
    if (lane == 0) wave_max[wave] = m;
    __syncthreads();
    if (tid == 0) {
        u64 mm = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) mm = wave_max[w] > mm ? wave_max[w] : mm;
        if (isV) lenV[0] = (long long)mm;
        else lenU[k] = (long long)mm;
    }
    __syncthreads();  // wave_max / wave_tot are reused by the workgroup's next column
  }
}

// DE/fdp.py:104-108: priorSum = sum(priors) + 0.01; priors[i] = priors[i]/priorSum + 0.01
constexpr int CLI_BLOCK = 512;  // the solver's set-up repeats this sum with the same workgroup shape
__global__ __launch_bounds__(CLI_BLOCK) void cli_init_kernel(const double* acc, long long K,
                                                             double* alpha0) {
    __shared__ double wsum[CLI_BLOCK / 64];
    const double priorSum = dme_cli_prior_sum<CLI_BLOCK>(acc, K, wsum);
    for (long long k = threadIdx.x; k < K; k += CLI_BLOCK)
        alpha0[k] = __dadd_rn(__ddiv_rn(acc[k], priorSum), 0.01);
}

}  // namespace

extern "C" int dme_suffix_scan(const uint64_t* hist, int64_t K, int64_t nbins,
                               const uint64_t* histV, int64_t tbins, int64_t* U, double* Uf,
                               int64_t ldU, int64_t* lenU, int64_t* V, double* Vf, int64_t* lenV,
                               void* stream) {
    DME_REQUIRE(K >= 1 && nbins >= 1 && tbins >= 1, "bad shape K=%lld nbins=%lld tbins=%lld",
                (long long)K, (long long)nbins, (long long)tbins);
    DME_REQUIRE(ldU >= nbins - 1, "ldU=%lld < nbins-1=%lld", (long long)ldU, (long long)nbins - 1);
    DME_REQUIRE(hist && histV && U && lenU && V && lenV, "null pointer");
    DME_REQUIRE((Uf == nullptr) == (Vf == nullptr), "Uf and Vf must both be given or both be NULL");
    hipLaunchKernelGGL(suffix_scan_kernel, dim3((unsigned)(K + 1)), dim3(SCAN_BLOCK), 0,
                       (hipStream_t)stream, (const u64*)hist, (long long)K, (long long)nbins,
                       (const u64*)histV, (long long)tbins, (long long*)U, Uf, (long long)ldU,
                       (long long*)lenU, (long long*)V, Vf, (long long*)lenV, (const long long*)nullptr,
                       (const long long*)nullptr, (int*)nullptr);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_suffix_scan_ragged(const uint64_t* compact, const int64_t* off, int64_t K,
                                      const uint64_t* histV, int64_t tbins, int64_t* U, double* Uf,
                                      int64_t* lenU, int64_t* V, double* Vf, int64_t* lenV,
                                      const int64_t* slotoff, int32_t* slotcol, void* stream) {
    DME_REQUIRE(K >= 1 && tbins >= 1, "bad shape K=%lld tbins=%lld", (long long)K, (long long)tbins);
    DME_REQUIRE(compact && off && histV && U && lenU && V && lenV, "null pointer");
    DME_REQUIRE((Uf == nullptr) == (Vf == nullptr), "Uf and Vf must both be given or both be NULL");
    DME_REQUIRE((slotoff == nullptr) == (slotcol == nullptr), "slotoff and slotcol go together");
    long long grid = K + 1;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(suffix_scan_kernel, dim3((unsigned)grid), dim3(SCAN_BLOCK), 0,
                       (hipStream_t)stream, (const u64*)compact, (long long)K, 0LL, (const u64*)histV,
                       (long long)tbins, (long long*)U, Uf, 0LL, (long long*)lenU, (long long*)V, Vf,
                       (long long*)lenV, (const long long*)off, (const long long*)slotoff,
                       (int*)slotcol);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_cli_init_alphas(const double* init_acc, int64_t K, double* alpha0,
                                   void* stream) {
    DME_REQUIRE(K >= 1 && init_acc && alpha0, "bad arguments");
    hipLaunchKernelGGL(cli_init_kernel, dim3(1), dim3(CLI_BLOCK), 0, (hipStream_t)stream, init_acc,
                       (long long)K, alpha0);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
