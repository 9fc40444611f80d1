// dme_weighted.hip — statistics of rows that carry a weight.
//
// CompressedRowData.appendRow(row, weight) (DirichletEstimation/dirichletMultinomialEstimation.py
// :113-123) adds `weight`, not 1, to U[k][j] (j < row[k]) and V[j] (j < sum(row)); the reference's
// Dirichlet-mixture EM calls it with fractional responsibilities
// (ConjugatePriorTools/dirichletMixtureEstimation.py:129-133).  In histogram form:
//     histW[k][v] += weight for x_ik == v,   histVW[t] += weight for sum_k x_ik == t      (fp64)
// and U/V are again reverse suffix sums.  len(U[k]) grows with every appended row whatever its
// weight, so the largest count per column / largest total are tracked as integers beside the sums.
// One wave per row.  Counts 1..B of every column and row totals below TB are privatised in LDS as
// fp64 bins (ds_add_f64; a wave's lanes own distinct columns, so the adds do not collide), the
// largest count per column as 32-bit LDS maxima; a workgroup adds its non-zero bins to the global
// histograms once, at the end.  Counts above B and long totals go straight to fp64 global atomics
// (they execute at the memory side, ~20 G/s chip-wide: the whole pass ran at that rate before the
// bins were privatised).  C weight columns per row (the C components of the mixture EM,
// CPT/dirichletMixtureEstimation.py:130-133) fill C histograms in the same pass over X.
// fp64 sums here depend on arrival order in the last bits (the reference adds row by row).
#include <type_traits>

#include "dme_common.h"

namespace {

constexpr int WBLOCK = 1024;
constexpr int WWAVES = WBLOCK / 64;
constexpr long long W_LDS_BYTES = 150 * 1024;

struct WArgs {
    const int* X;
    const double* wts;  // [N][C]
    long long N, K, ld, C;
    double* histW;   // [C][K][nbins]
    long long nbins;
    double* histVW;  // [C][tbins]
    long long tbins;
    long long* maxcol;
    long long* maxtot;
    u64* status;
    int KP;      // doubles from one LDS bin row to the next: K rounded up to a multiple of 32 (256 bytes =
                 // all LDS banks), so that a lane's bank depends on its column alone - a wave's lanes,
                 // which own distinct consecutive columns, then hit the banks like a linear access
                 // whatever their counts (ds_add_f64: 8 cycles per wave instruction against 25 for
                 // random addresses, profiles/r3_lds_atomics_microbench.txt)
    int B;       // counts 1..B live in LDS
    int TB;      // totals below TB live in LDS
    int mxlds;   // column maxima in LDS (K small enough)
    long long rows_per_wg;
    double* img;           // NULL, or one image of the LDS bins per workgroup: [grid][img_stride]
    long long img_stride;  // doubles from one image to the next (>= C * TB + C * B * KP)
};

// J > 0: a lane keeps its J cells of a row (columns lane + 64 j) in registers and the next row's
// loads are in flight while the current row is processed; J == 0 (K > 1024): the row is read twice.
// CC > 0: the number of weight columns is known at compile time (1 ... 8: a mixture's components), so
// the C adds of a cell are straight-line code - one address, C ds_add_f64 at fixed distances - instead of
// a scalar loop with a branch per weight (108 scalar instructions per row of K = 100 before).
template <int J, int CC>
__global__ __launch_bounds__(WBLOCK) void stats_weighted_kernel(WArgs a) {
    extern __shared__ double wl[];
    const long long K = a.K, C = CC > 0 ? CC : a.C;
    const int B = a.B, TB = a.TB;
    double* hv = wl;                              // [C][TB]
    const long long KP = a.KP;
    double* h = hv + C * TB;                      // [C][B][KP]: count v of column k at ((c*B + v-1)*KP + k)
    u32* mx = (u32*)(h + C * (long long)B * KP);  // [K] when mxlds
    __shared__ u64 sh_stat[8];  // [1] cells >= nbins [2] totals >= tbins [3] negatives [5] max total
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long nd = C * TB + C * (long long)B * KP;
    for (long long i = tid; i < nd; i += WBLOCK) wl[i] = 0.0;
    if (a.mxlds)
        for (long long k = tid; k < K; k += WBLOCK) mx[k] = 0;
    if (tid < 8) sh_stat[tid] = 0;
    __syncthreads();
    const long long row0 = (long long)blockIdx.x * a.rows_per_wg;
    const long long row1 = row0 + a.rows_per_wg < a.N ? row0 + a.rows_per_wg : a.N;
    u64 bad_cells = 0, bad_rows = 0, neg = 0;
    constexpr int JR = J > 0 ? J : 1;
    // Rows in flight per wave: a row of K = 100 int32 cells is 400 bytes, and with one row ahead the
    // whole chip had 1.6 MB outstanding - the pass waited for memory latency, once for the row and
    // once more for its weights, which were fetched when first used.  DEPTH rows and their weights
    // (lane c holds weight c of the row) are loaded ahead now.
    constexpr int DEPTH = J > 0 ? (J <= 4 ? 4 : 2) : 1;
    int rowbuf[DEPTH][JR];
    double wbuf[DEPTH];
    auto load_row = [&](int (&d)[JR], double& wv, long long i) {
        if constexpr (J > 0) {
            const int* row = a.X + i * a.ld;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const long long k = lane + 64 * j;
                d[j] = (i < row1 && k < K) ? __builtin_nontemporal_load(row + k) : 0;
            }
            wv = (i < row1 && lane < C) ? a.wts[i * C + lane] : 0.0;
        }
    };
    // weight c of the row in hand: from the lane that holds it (wv), or from memory (J == 0)
    struct Wts {
        const double* p;
        double v;
        __device__ __forceinline__ double operator[](long long c) const {
            return p ? p[c] : dme_readlane_f64(v, (int)c);
        }
    };
    // The row-total bins take the C weights in ONE LDS instruction: lane c adds weight c (the lane that
    // holds it anyway) - five single-lane adds before.  The largest total is wave-uniform: a register.
    u64 maxt_seen = 0;
    auto total_of = [&](u64 total, const Wts& w) {
        if (total > 0) {  // wave-uniform
            if ((long long)total < a.tbins) {
                maxt_seen = total > maxt_seen ? total : maxt_seen;
                if (lane < C) {
                    const double wc = w.p ? w.p[lane] : w.v;
                    if (total < (u64)TB) atomicAdd(&hv[lane * TB + (long long)total], wc);
                    else atomicAdd(&a.histVW[lane * a.tbins + (long long)total], wc);
                }
            } else if (lane == 0) {
                bad_rows++;
            }
        }
    };
    // J > 0: a lane's columns are fixed (lane + 64 j), so their largest counts live in registers and
    // reach LDS once, at the end - an LDS atomic max per non-zero cell before.
    u32 mxr[JR];
#pragma unroll
    for (int j = 0; j < JR; ++j) mxr[j] = 0;
    // A cell is classified first (zero / negative / beyond the histogram / counted: its column maximum
    // noted), then its C weights are added.  The weights reach the adds as wave-uniform scalars that
    // the caller read off their lanes in UNIFORM control flow: reading them behind the divergent
    // `v == 0` test went wrong once registers ran short - a value reloaded from scratch under a partial
    // exec mask is stale in the lanes that are off, and v_readlane reads those lanes too.
    auto classify = [&](int v, long long k, u32* mreg) -> bool {
        if (v == 0) return false;
        if (v < 0) { neg++; return false; }
        if ((long long)v >= a.nbins) { bad_cells++; return false; }
        if (mreg) {
            *mreg = (u32)v > *mreg ? (u32)v : *mreg;
        } else if (a.mxlds) {
            atomicMax(&mx[k], (u32)v);
        } else if (__builtin_nontemporal_load(&a.maxcol[k]) < (long long)v) {
            atomicMax((u64*)&a.maxcol[k], (u64)v);  // the maxima settle after a few rows
        }
        return true;
    };
    auto add_one = [&](int v, long long k, long long c, double wc) {
        // (the LDS add through an explicit LDS pointer: left as two generic atomics, the compiler folded
        // the branches into one flat atomic on a selected pointer and its address-space test did not
        // assemble - "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
        typedef __attribute__((address_space(3))) double lds_double;
        if (v <= B)
            __hip_atomic_fetch_add((lds_double*)&h[(c * B + (v - 1)) * KP + k], wc, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            atomicAdd(&a.histW[(c * K + k) * a.nbins + v], wc);
    };
    auto add_fixed = [&](int v, long long k, const double* wsc) {  // CC > 0: straight-line code
        if (v <= B) {
            double* cellp = &h[(long long)(v - 1) * KP + k];
            const long long cstride = (long long)B * KP;
#pragma unroll
            for (int c = 0; c < CC; ++c)
                __hip_atomic_fetch_add((__attribute__((address_space(3))) double*)(cellp + c * cstride), wsc[c],
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
#pragma unroll
            for (int c = 0; c < CC; ++c) atomicAdd(&a.histW[((long long)c * K + k) * a.nbins + v], wsc[c]);
        }
    };
    if constexpr (J > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load_row(rowbuf[d], wbuf[d], row0 + wave + (long long)d * WWAVES);
    }
    for (long long i0 = row0 + wave; i0 < row1; i0 += (long long)WWAVES * DEPTH) {
        if constexpr (J > 0) {
            // The DEPTH rows in hand first give up their cells and row totals - DEPTH independent DPP
            // reductions side by side - and their buffers are refilled; only then do the LDS adds of the
            // rows follow one another (with one row at a time the wave sat through a reduction's whole
            // dependent chain before every row's adds: 4 waves per SIMD do not hide that).
            int cur[DEPTH][JR];
            double wcur[DEPTH];
            u64 tot[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const long long i = i0 + (long long)d * WWAVES;
#pragma unroll
                for (int j = 0; j < J; ++j) cur[d][j] = rowbuf[d][j];
                wcur[d] = wbuf[d];
                load_row(rowbuf[d], wbuf[d], i + (long long)WWAVES * DEPTH);
            }
            u32 m = 0;
            u32 s32[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                s32[d] = 0;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const u32 pos = cur[d][j] > 0 ? (u32)cur[d][j] : 0u;
                    m |= pos;
                    s32[d] += pos;
                }
            }
            // 64 lanes x J cells below 2^20 cannot wrap 32 bits: the DPP sums
            if (__builtin_expect(!__any(m >= (1u << 20)), 1)) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) tot[d] = (u64)dme_wave_sum_u32(s32[d]);
            } else {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    u64 s = 0;
#pragma unroll
                    for (int j = 0; j < J; ++j) s += (u64)(cur[d][j] > 0 ? cur[d][j] : 0);
                    tot[d] = dme_wave_sum_u64(s);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const long long i = i0 + (long long)d * WWAVES;
                if (i < row1) {  // wave-uniform
                    const Wts w{nullptr, wcur[d]};
                    total_of(tot[d], w);
                    bool ok[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) ok[j] = classify(cur[d][j], lane + 64 * j, &mxr[j]);
                    if constexpr (CC > 0) {
                        double wsc[CC];
#pragma unroll
                        for (int c = 0; c < CC; ++c) wsc[c] = dme_readlane_f64(wcur[d], c);
#pragma unroll
                        for (int j = 0; j < J; ++j)
                            if (ok[j]) add_fixed(cur[d][j], lane + 64 * j, wsc);
                    } else {
                        for (long long c = 0; c < C; ++c) {
                            const double wc = dme_readlane_f64(wcur[d], __builtin_amdgcn_readfirstlane((int)c));
#pragma unroll
                            for (int j = 0; j < J; ++j)
                                if (ok[j]) add_one(cur[d][j], lane + 64 * j, c, wc);
                        }
                    }
                }
            }
        } else {
            const long long i = i0;
            const Wts w{a.wts + i * C, 0.0};
            const int* row = a.X + i * a.ld;
            u64 s = 0;
            for (long long k = lane; k < K; k += 64) {
                const int v = row[k];
                s += (u64)(v > 0 ? v : 0);
            }
            total_of(dme_wave_sum_u64(s), w);
            for (long long k = lane; k < K; k += 64) {  // second read: L1 / L2
                const int v = row[k];
                if (classify(v, k, nullptr))
                    for (long long c = 0; c < C; ++c) add_one(v, k, c, w.p[c]);  // (weights from memory)
            }
        }
    }
    bad_cells = dme_wave_sum_u64(bad_cells);
    bad_rows = dme_wave_sum_u64(bad_rows);
    neg = dme_wave_sum_u64(neg);
    if constexpr (J > 0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const long long k = lane + 64 * j;
            if (mxr[j] && k < K) {
                if (a.mxlds) atomicMax(&mx[k], mxr[j]);
                else if (a.maxcol[k] < (long long)mxr[j]) atomicMax((u64*)&a.maxcol[k], (u64)mxr[j]);
            }
        }
    }
    if (lane == 0 && maxt_seen) atomicMax(&sh_stat[5], maxt_seen);
    if (lane == 0) {
        if (bad_cells) atomicAdd(&sh_stat[1], bad_cells);
        if (bad_rows) atomicAdd(&sh_stat[2], bad_rows);
        if (neg) atomicAdd(&sh_stat[3], neg);
    }
    __syncthreads();
    const long long BK = (long long)B * KP;
    if (a.img) {
        // The workgroup's bins leave as one image (consecutive lanes, consecutive doubles) and
        // weighted_reduce_kernel adds the images in a fixed order: 256 workgroups that each sent their
        // non-zero bins to the SAME few thousand addresses with fp64 global atomics spent 0.10 ms there,
        // whatever the number of rows (1 M x 100 x 5 weights: 0.37 ms in all).
        double* out = a.img + (long long)blockIdx.x * a.img_stride;
        for (long long i = tid; i < nd; i += WBLOCK) __builtin_nontemporal_store(wl[i], out + i);
    } else {
    // the workgroup's bins -> the global histograms (zero bins cost nothing)
    for (long long i = tid; i < C * BK; i += WBLOCK) {
        const double x = h[i];
        if (x != 0.0) {  // (the padding columns k >= K stay zero)
            const long long c = i / BK, rem = i - c * BK;
            const long long vb = rem / KP, k = rem - vb * KP;
            atomicAdd(&a.histW[(c * K + k) * a.nbins + vb + 1], x);
        }
    }
    for (long long i = tid; i < C * TB; i += WBLOCK) {
        const double x = hv[i];
        if (x != 0.0) {
            const long long c = i / TB, t = i - c * TB;
            atomicAdd(&a.histVW[c * a.tbins + t], x);
        }
    }
    }
    if (a.mxlds)
        for (long long k = tid; k < K; k += WBLOCK) {
            const u64 m = mx[k];
            if (m && (u64)a.maxcol[k] < m) atomicMax((u64*)&a.maxcol[k], m);
        }
    if (tid == 0) {
        if (row1 > row0) atomicAdd(&a.status[0], (u64)(row1 - row0));
        if (sh_stat[5] && (u64)a.maxtot[0] < sh_stat[5]) atomicMax((u64*)a.maxtot, sh_stat[5]);
    }
    if (tid >= 1 && tid <= 3 && sh_stat[tid]) atomicAdd(&a.status[tid], sh_stat[tid]);
}

// Adds the workgroups' images into the global histograms: a workgroup takes 64 consecutive bins, its four
// waves a quarter of the images each (in index order, sixteen loads in flight), wave 0 adds the four
// partial sums in wave order and is the only writer of its bins - plain `+=`, and the same bits from run
// to run for the same images.
constexpr int RBLOCK = 256;
__global__ __launch_bounds__(RBLOCK) void weighted_reduce_kernel(WArgs a, int nimg) {
    __shared__ double part[RBLOCK / 64][64];
    const long long C = a.C, TB = a.TB, KP = a.KP, K = a.K;
    const long long BK = (long long)a.B * KP;
    const long long nd = C * TB + C * BK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    const int per = (nimg + RBLOCK / 64 - 1) / (RBLOCK / 64);
    const int g0 = wave * per, g1 = g0 + per < nimg ? g0 + per : nimg;
    double s = 0.0;
    if (i < nd) {
        const double* p = a.img + i;
        int g = g0;
        for (; g + 16 <= g1; g += 16) {
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = __builtin_nontemporal_load(p + (long long)(g + u) * a.img_stride);
#pragma unroll
            for (int u = 0; u < 16; ++u) s = __dadd_rn(s, x[u]);
        }
        for (; g < g1; ++g) s = __dadd_rn(s, __builtin_nontemporal_load(p + (long long)g * a.img_stride));
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || i >= nd) return;
    double x = part[0][lane];
#pragma unroll
    for (int w = 1; w < RBLOCK / 64; ++w) x = __dadd_rn(x, part[w][lane]);
    if (x == 0.0) return;  // (the padding columns k >= K stay zero)
    if (i < C * TB) {
        const long long c = i / TB, t = i - c * TB;
        double* d = &a.histVW[c * a.tbins + t];
        *d = __dadd_rn(*d, x);
    } else {
        const long long r = i - C * TB;
        const long long c = r / BK, rem = r - c * BK;
        const long long vb = rem / KP, k = rem - vb * KP;
        double* d = &a.histW[(c * K + k) * a.nbins + vb + 1];
        *d = __dadd_rn(*d, x);
    }
}

// Uf[k][n] += sum_{v > n} histW[k][v]; lenU[k] = max(lenU[k], maxcol[k]).  One thread per column
// walks the bins from the top (the weighted statistics are small); column K is V.
__global__ __launch_bounds__(256) void suffix_scan_weighted_kernel(
    const double* histW, long long K, long long nbins, const double* histVW, long long tbins,
    const long long* maxcol, const long long* maxtot, double* Uf, long long ldU, long long* lenU,
    double* Vf, long long capV, long long* lenV, long long sU, long long sLen) {
    // blockIdx.y = weight column c (dme_suffix_scan_weighted_batch): its histograms lie K * nbins / tbins
    // doubles further on, its U / V / lengths sU / capV / sLen / 1 elements
    {
        const long long c = blockIdx.y;
        histW += c * K * nbins;
        histVW += c * tbins;
        Uf += c * sU;
        lenU += c * sLen;
        Vf += c * capV;
        lenV += c;
    }
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

namespace {

// How a workgroup's 150 KB of LDS are shared out: column maxima (4 B each) when they fit in a quarter of
// the budget, then row-total bins (at most a quarter), the rest for count bins 1..B of every column and
// weight column.
struct WPlan {
    int mxlds;
    long long TB, KP, B;
    long long nd;  // doubles of a workgroup's image: C * TB + C * B * KP
};
WPlan plan_weighted(long long K, long long C, long long nbins, long long tbins) {
    WPlan p{};
    long long budget = W_LDS_BYTES;
    p.mxlds = K * 4 <= budget / 4 ? 1 : 0;
    if (p.mxlds) budget -= (K * 4 + 7) & ~7LL;
    long long TB = tbins;
    if (TB * C * 8 > budget / 4) TB = budget / 4 / (C * 8);
    p.TB = TB;
    budget -= TB * C * 8;
    // (only the pitch matters: the base of the bins shifts every lane's bank alike)
    p.KP = (K + 31) & ~31LL;
    long long B = budget / (C * p.KP * 8);
    if (B > nbins - 1) B = nbins - 1;
    if (B < 0) B = 0;
    p.B = B;
    p.nd = C * TB + C * B * p.KP;
    return p;
}
// Workgroups from which on the bins leave as images for weighted_reduce_kernel (fewer: their atomics are
// cheaper than a second launch).
constexpr long long IMG_MIN_GRID = 32;

int stats_weighted(const int32_t* X, const double* weights, int64_t C, int64_t N, int64_t K,
                   int64_t ld, double* histW, int64_t nbins, double* histVW, int64_t tbins,
                   int64_t* maxcol, int64_t* maxtot, uint64_t* status, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    DME_REQUIRE(N >= 0 && K >= 1 && ld >= K, "bad shape N=%lld K=%lld ld=%lld", (long long)N,
                (long long)K, (long long)ld);
    DME_REQUIRE(C >= 1 && C <= 64, "weight columns per row: 1..64, got %lld", (long long)C);
    DME_REQUIRE(nbins >= 1 && tbins >= 1, "nbins/tbins must be >= 1");
    DME_REQUIRE(nbins <= 0xFFFFFFFFLL, "nbins too large");
    DME_REQUIRE(histW && histVW && maxcol && maxtot && status, "null output pointer");
    DME_REQUIRE(workspace_bytes >= 0 && (workspace || workspace_bytes == 0), "bad workspace");
    DME_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    if (N == 0) return DME_OK;
    DME_REQUIRE(X && weights, "null input pointer");
    WArgs a{};
    a.X = X; a.wts = weights; a.N = N; a.K = K; a.ld = ld; a.C = C;
    a.histW = histW; a.nbins = nbins; a.histVW = histVW; a.tbins = tbins;
    a.maxcol = (long long*)maxcol; a.maxtot = (long long*)maxtot; a.status = (u64*)status;
    const WPlan pl = plan_weighted(K, C, nbins, tbins);
    a.mxlds = pl.mxlds;
    const long long TB = pl.TB, KP = pl.KP, B = pl.B;
    a.TB = (int)TB;
    a.KP = (int)KP;
    a.B = (int)B;
    const size_t lds = (size_t)(C * TB * 8 + C * B * KP * 8 + (a.mxlds ? ((K * 4 + 7) & ~7LL) : 0));
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    // a workgroup should see enough rows to repay the flush of its bins
    long long grid = (N + 255) / 256;
    if (grid > cus) grid = cus;
    if (grid < 1) grid = 1;
    a.rows_per_wg = (N + grid - 1) / grid;
    grid = (N + a.rows_per_wg - 1) / a.rows_per_wg;
    const bool images = grid >= IMG_MIN_GRID && pl.nd > 0 && workspace_bytes >= grid * pl.nd * 8;
    if (images) {
        a.img = (double*)workspace;
        a.img_stride = pl.nd;
    }
    auto launch = [&](auto kern) -> int {
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WBLOCK), lds, (hipStream_t)stream, a);
        return DME_OK;
    };
    auto by_c = [&](auto jtag) -> int {
        constexpr int J = decltype(jtag)::value;
        switch ((int)C) {
            case 1: return launch(stats_weighted_kernel<J, 1>);
            case 2: return launch(stats_weighted_kernel<J, 2>);
            case 3: return launch(stats_weighted_kernel<J, 3>);
            case 4: return launch(stats_weighted_kernel<J, 4>);
            case 5: return launch(stats_weighted_kernel<J, 5>);
            case 6: return launch(stats_weighted_kernel<J, 6>);
            case 7: return launch(stats_weighted_kernel<J, 7>);
            case 8: return launch(stats_weighted_kernel<J, 8>);
            default: return launch(stats_weighted_kernel<J, 0>);
        }
    };
    int rc;
    if (K <= 128) rc = by_c(std::integral_constant<int, 2>{});
    else if (K <= 256) rc = by_c(std::integral_constant<int, 4>{});
    else if (K <= 1024) rc = by_c(std::integral_constant<int, 16>{});
    else rc = by_c(std::integral_constant<int, 0>{});
    if (rc) return rc;
    if (images)
        hipLaunchKernelGGL(weighted_reduce_kernel, dim3((unsigned)((pl.nd + 63) / 64)), dim3(RBLOCK), 0,
                           (hipStream_t)stream, a, (int)grid);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

}  // namespace

extern "C" int64_t dme_stats_weighted_workspace_bytes(int64_t K, int64_t C, int64_t nbins, int64_t tbins) {
    if (K < 1 || C < 1 || C > 64 || nbins < 1 || tbins < 1) return 0;
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    return (int64_t)cus * plan_weighted(K, C, nbins, tbins).nd * 8;
}

extern "C" int dme_stats_dense_weighted_batch_ws_i32(const int32_t* X, const double* weights, int64_t C,
                                                     int64_t N, int64_t K, int64_t ld, double* histW,
                                                     int64_t nbins, double* histVW, int64_t tbins,
                                                     int64_t* maxcol, int64_t* maxtot, uint64_t* status,
                                                     void* workspace, int64_t workspace_bytes,
                                                     void* stream) {
    return stats_weighted(X, weights, C, N, K, ld, histW, nbins, histVW, tbins, maxcol, maxtot, status,
                          workspace, workspace_bytes, stream);
}

extern "C" int dme_stats_dense_weighted_i32(const int32_t* X, const double* weights, int64_t N,
                                            int64_t K, int64_t ld, double* histW, int64_t nbins,
                                            double* histVW, int64_t tbins, int64_t* maxcol,
                                            int64_t* maxtot, uint64_t* status, void* stream) {
    return stats_weighted(X, weights, 1, N, K, ld, histW, nbins, histVW, tbins, maxcol, maxtot,
                          status, nullptr, 0, stream);
}

extern "C" int dme_stats_dense_weighted_batch_i32(const int32_t* X, const double* weights,
                                                  int64_t C, int64_t N, int64_t K, int64_t ld,
                                                  double* histW, int64_t nbins, double* histVW,
                                                  int64_t tbins, int64_t* maxcol, int64_t* maxtot,
                                                  uint64_t* status, void* stream) {
    return stats_weighted(X, weights, C, N, K, ld, histW, nbins, histVW, tbins, maxcol, maxtot,
                          status, nullptr, 0, stream);
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
                       (long long)ldU, (long long*)lenU, Vf, (long long)capV, (long long*)lenV, 0LL, 0LL);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_suffix_scan_weighted_batch(const double* histW, int64_t C, int64_t K, int64_t nbins,
                                              const double* histVW, int64_t tbins, const int64_t* maxcol,
                                              const int64_t* maxtot, double* Uf, int64_t ldU,
                                              int64_t* lenU, double* Vf, int64_t capV, int64_t* lenV,
                                              void* stream) {
    DME_REQUIRE(C >= 1 && C <= 64 && K >= 1 && nbins >= 1 && tbins >= 1, "bad shape");
    DME_REQUIRE(ldU >= nbins - 1 && capV >= tbins - 1, "U/V too short for the histograms");
    DME_REQUIRE(histW && histVW && maxcol && maxtot && Uf && lenU && Vf && lenV, "null pointer");
    hipLaunchKernelGGL(suffix_scan_weighted_kernel, dim3((unsigned)((K + 1 + 255) / 256), (unsigned)C),
                       dim3(256), 0, (hipStream_t)stream, histW, (long long)K, (long long)nbins, histVW,
                       (long long)tbins, (const long long*)maxcol, (const long long*)maxtot, Uf,
                       (long long)ldU, (long long*)lenU, Vf, (long long)capV, (long long*)lenV,
                       (long long)(K * ldU), (long long)K);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}
