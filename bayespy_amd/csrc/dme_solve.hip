// dme_solve.hip — the Newton-Raphson solver of the Dirichlet-multinomial MLE on MI355X (gfx950).
//
// Replaces DirichletEstimation/dirichletMultinomialEstimation.py ("DE/dme.py"):
//   dirichLogProb :16-29, priorGradient :32-48, priorHessianConst :52-53, priorHessianDiag :55-61,
//   getPredictedStep :67-72, getPredictedStepAlt :76-80, testTrialPriors :97-102,
//   sqVectorSize :104, findDirichletPriors :126-190.
//
// Structure.  The ragged statistics are cut into segments of SEG consecutive n of one column (the
// row-total vector V is column K, evaluated at A = sum(alpha)); a segment's slot in the partial-sum
// table follows from its column and start (slots_per_column).  One evaluation is
//   eval_kernel  — one wave per segment: the fp64 sums  sum U*log(a+n),  sum U/(a+n),
//                  sum U/(a+n)^2  with compensated (two-sum) accumulation and DPP wave reductions;
//   step_kernel  — one workgroup: gathers the per-column sums in a fixed order, forms loss,
//                  gradient and the diagonal-plus-rank-one Hessian, and advances the reference's
//                  state machine (strict-< Newton acceptance, log-space trial, x0.9 step halving,
//                  exits A/B/C) — every decision is taken here, on the device.
// A loss evaluation at a trial point also yields gradient and Hessian there, so an accepted Newton
// step costs exactly one evaluation.  The host enqueues batches of (eval, step) pairs and polls a
// done flag once per batch; kernels after convergence return at once.  A kernel boundary
// (~1.5 us on MI355X) is cheaper than an in-kernel grid barrier (~4-7 us), which is why the loop is
// a chain of launches and not one persistent kernel.  No MFMA: nothing here is a dense contraction.
#include <math.h>
#include <stdlib.h>

#include <mutex>

#include <atomic>

#include "dme_common.h"

namespace {

constexpr int SEG = 256;         // terms per segment: 4 per lane
static_assert(SEG == DME_SEGMENT_TERMS, "dme_ragged_plan (dme_shard.hip) cuts columns into slots of this size");
#ifndef DME_STEP_BLOCK
#define DME_STEP_BLOCK 512  // 1024-thread single-workgroup kernels measured 2x slower per launch
#endif
constexpr int STEP_BLOCK = DME_STEP_BLOCK;
constexpr int STEP_WAVES = STEP_BLOCK / 64;
constexpr long long SMALL_NSEG = 64;  // at most this many segments: the single-launch solver
constexpr int EVAL_BLOCK = 256;

enum Phase : long long { PH_INIT = 0, PH_NEWTON = 1, PH_LOGSPACE = 2, PH_HALVING = 3, PH_EVALONLY = 4 };

struct Ctrl {
    long long done;
    long long phase;
    long long count;
    long long n_evals;
    long long exit_reason;
    long long error;
    long long nseg;
    long long launches;
    long long trial_valid;
    long long branch;
    double currentLoss;
    double learnRate;
    double A_trial;
    double hconst_cur;
    long long fragile;  // accept / reject decisions taken on |trial loss - current loss| <= ~4 ulp
    double pad;
};
static_assert(sizeof(Ctrl) % 16 == 0, "Ctrl must stay 16-byte sized");

struct Partial {  // per segment
    double s0hi, s0lo, s1hi, s1lo, s2, pad;
};

// Prior terms on top of the Dirichlet-multinomial likelihood.
//   lambda : the scalar hyperprior of DirichletEstimation (DE/dme.py:28 loss += lambda*A, :46
//            gradient += lambda*A — the reference's own quirk, kept).
//   beta,W : the (Beta, W) conjugate hyperprior of ConjugatePriorTools/dirichletMultinomialEstimation.py
//            ("CPT/dme.py"): loss -= sum alpha_k Beta_k (:32-34), += W*(lgamma(A) - sum lgamma(alpha_k))
//            (:36-38); gradient -= Beta_k (:58-60), += W*(psi(A) - psi(alpha_k)) (:62-64);
//            hConst += W*psi'(A) (:74-75); hDiag -= W*psi'(alpha_k) (:84-86).
//   loss_ignores_prior : CPT's getTotalLoss drops Beta and W (CPT/dme.py:109-110), so inside
//            findDirichletPriors the loss has no prior term while gradient and Hessian do.
struct Prior {
    double lambda;
    const double* beta;
    double W;
    int has_W;
    int loss_ignores_prior;
};

// digamma / trigamma for x > 0: recurrence up to x >= 10, then the asymptotic series (next omitted
// term < 1e-16).  The reference calls scipy.special.psi / polygamma(1, .) (CPT/dme.py:15-16).
__device__ double dme_digamma(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
                     f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x + t;
}

__device__ double dme_trigamma(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r += 1.0 / (x * x);
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = (1.0 / x) * f * (1.0 / 6.0 + f * (-1.0 / 30.0 + f * (1.0 / 42.0 + f * (-1.0 / 30.0 +
                     f * (5.0 / 66.0 + f * (-691.0 / 2730.0 + f * (7.0 / 6.0)))))));
    return r + 1.0 / x + 0.5 * f + t;
}

// workspace carve-up (all offsets 256-byte aligned)
struct Ws {
    Ctrl* ctrl;
    Partial* part;  // [K * S + SV] partial sums, one slot per segment (see slots_per_column)
    double* alpha_cur;
    double* g_cur;
    double* d_cur;
    double* alpha_trial;
    double* g_trial;
    double* d_trial;
    // second copies of the state that crosses a launch, for the one-launch-per-evaluation path
    // (iter_kernel): launch j reads copy j & 1 and writes the other
    Ctrl* ctrl2;
    Partial* part2;
    double* alpha_cur2;
    double* g_cur2;
    double* d_cur2;
    double* alpha_trial2;
    long long nseg_max;  // = K * S + SV slots
    long long bytes;
};

inline long long align256(long long x) { return (x + 255) & ~255LL; }

// Segments are SEG consecutive n of one column.  Column k owns the S = ceil(ldU / SEG) slots
// [k * S, (k + 1) * S) of the partial-sum table and V (column K) the SV = ceil(capV / SEG) slots behind
// them: a slot's column and start follow from its index and a column's slots from lenU[k] alone, so
// neither the evaluation nor the gather chases a segment table through memory (two dependent loads
// per launch less on the solver's critical path).  Slots beyond a column's length are never written
// nor read.
__host__ __device__ inline long long slots_per_column(long long ld) {
    const long long n = (ld + SEG - 1) / SEG;
    return n < 1 ? 1 : n;
}

// nU: the slots of the K columns - K * slots_per_column(ldU) for dense-padded U, the ragged layout's own
// count otherwise.
Ws carve(void* base, long long K, long long nU, long long capV) {
    Ws w{};
    long long off = 0;
    char* b = (char*)base;
    auto take = [&](long long bytes) {
        char* p = b ? b + off : nullptr;
        off += align256(bytes);
        return p;
    };
    w.nseg_max = nU + slots_per_column(capV);
    w.ctrl = (Ctrl*)take(sizeof(Ctrl));
    w.part = (Partial*)take(w.nseg_max * (long long)sizeof(Partial));
    w.alpha_cur = (double*)take(K * 8);
    w.g_cur = (double*)take(K * 8);
    w.d_cur = (double*)take(K * 8);
    w.alpha_trial = (double*)take(K * 8);
    w.g_trial = (double*)take(K * 8);
    w.d_trial = (double*)take(K * 8);
    w.ctrl2 = (Ctrl*)take(sizeof(Ctrl));
    w.part2 = (Partial*)take(w.nseg_max * (long long)sizeof(Partial));
    w.alpha_cur2 = (double*)take(K * 8);
    w.g_cur2 = (double*)take(K * 8);
    w.d_cur2 = (double*)take(K * 8);
    w.alpha_trial2 = (double*)take(K * 8);
    w.bytes = off;
    return w;
}

// ---------------------------------------------------------------------------------------------
// block-wide reductions (STEP_BLOCK threads), deterministic order
// ---------------------------------------------------------------------------------------------
struct BlockShared {
    DmeAcc acc[STEP_WAVES];
    double dbl[STEP_WAVES];
    int flag[STEP_WAVES];
    // fused reduction (block_reduce_fused): up to 3 compensated sums, 6 plain sums, one flag
    DmeAcc facc[3][STEP_WAVES];
    double fdbl[6][STEP_WAVES];
    int fflag[STEP_WAVES];
};

// One pass, two barriers: NA compensated sums, ND plain sums and an OR-flag reduced together, in a
// fixed order (DPP inside a wave, then waves 0..15 in order).  Every thread gets the results.
template <int NA, int ND>
__device__ __forceinline__ void block_reduce_fused(DmeAcc (&a)[NA], double (&d)[ND], bool& flag,
                                                   BlockShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = dme_wave_sum_acc(a[i]);
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = dme_wave_sum_f64(d[i]);
    const int any = __any(flag);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) sh.facc[i][wave] = a[i];
#pragma unroll
        for (int i = 0; i < ND; ++i) sh.fdbl[i][wave] = d[i];
        sh.fflag[wave] = any;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        DmeAcc r = sh.facc[i][0];
#pragma unroll
        for (int w = 1; w < STEP_WAVES; ++w) r.add(sh.facc[i][w]);
        a[i] = r;
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        double r = sh.fdbl[i][0];
#pragma unroll
        for (int w = 1; w < STEP_WAVES; ++w) r = __dadd_rn(r, sh.fdbl[i][w]);
        d[i] = r;
    }
    int f = 0;
#pragma unroll
    for (int w = 0; w < STEP_WAVES; ++w) f |= sh.fflag[w];
    flag = f != 0;
    __syncthreads();
}

// The same without compensated sums (same order of additions as block_reduce_fused's plain sums).
template <int ND>
__device__ __forceinline__ void block_reduce_plain(double (&d)[ND], bool& flag, BlockShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = dme_wave_sum_f64(d[i]);
    const int any = __any(flag);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < ND; ++i) sh.fdbl[i][wave] = d[i];
        sh.fflag[wave] = any;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        double r = sh.fdbl[i][0];
#pragma unroll
        for (int w = 1; w < STEP_WAVES; ++w) r = __dadd_rn(r, sh.fdbl[i][w]);
        d[i] = r;
    }
    int f = 0;
#pragma unroll
    for (int w = 0; w < STEP_WAVES; ++w) f |= sh.fflag[w];
    flag = f != 0;
    __syncthreads();
}

// Block minimum, returned to every thread (used off the hot path only).
__device__ double block_min_f64(double x, BlockShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) x = fmin(x, __shfl_xor(x, d));
    if (lane == 0) sh.fdbl[0][wave] = x;
    __syncthreads();
    double r = sh.fdbl[0][0];
#pragma unroll
    for (int w = 1; w < STEP_WAVES; ++w) r = fmin(r, sh.fdbl[0][w]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ DmeAcc block_sum_acc(DmeAcc v, BlockShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = dme_wave_sum_acc(v);
    if (lane == 0) sh.acc[wave] = v;
    __syncthreads();
    DmeAcc r = sh.acc[0];
#pragma unroll
    for (int w = 1; w < STEP_WAVES; ++w) r.add(sh.acc[w]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_sum(double v, BlockShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = dme_wave_sum_f64(v);
    if (lane == 0) sh.dbl[wave] = v;
    __syncthreads();
    double r = sh.dbl[0];
#pragma unroll
    for (int w = 1; w < STEP_WAVES; ++w) r = __dadd_rn(r, sh.dbl[w]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ bool block_any(bool p, BlockShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int any = __any(p);
    if (lane == 0) sh.flag[wave] = any;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < STEP_WAVES; ++w) r |= sh.flag[w];
    __syncthreads();
    return r != 0;
}

// ---------------------------------------------------------------------------------------------
// evaluation: one wave per segment
// ---------------------------------------------------------------------------------------------
struct EvalArgs {  // the statistics as the entry points receive them
    const double* Uf;
    long long ldU;
    const long long* lenU;
    const double* Vf;
    long long capV;
    const long long* lenV;
    // RAGGED layout (offU != NULL; include/dme_hip.h, dme_ragged_t): column k's terms start at
    // Uf[offU[k]] and it owns the slots [slotoff[k], slotoff[k + 1]); slotcol[s] names the column of
    // slot s.  A column then takes the memory and the slots of its own length, not of the longest
    // column's (DE/dme.py:118: the reference's U[k] are lists of their own lengths).
    const long long* offU;
    const long long* slotoff;
    const int* slotcol;
    long long nslotsU;
};

// slots of the K columns (V's slots follow them)
__device__ __forceinline__ long long eval_nslots_u(const EvalArgs& e, long long K, long long S) {
    return e.offU ? e.nslotsU : K * S;
}
// first slot of column k
__device__ __forceinline__ long long eval_slot_base(const EvalArgs& e, long long k, long long S) {
    return e.offU ? e.slotoff[k] : k * S;
}
// len(U[k]), clamped to the column's allocation
__device__ __forceinline__ long long eval_col_len(const EvalArgs& e, long long k) {
    const long long L = e.lenU[k];
    const long long cap = e.offU ? e.offU[k + 1] - e.offU[k] : e.ldU;
    return L > cap ? cap : L;
}

// What a wave can load for its first segment before the trial point is known (iter_kernel issues
// these loads at kernel start, beside the loads of the state): the column's length and the wave's
// four terms per lane.
struct EvalPrefetch {
    long long L;
    double u[SEG / 64];
};

__device__ __forceinline__ void eval_slot_of(const EvalArgs& e, long long K, long long S, long long s,
                                             long long& k, long long& st, const double*& u,
                                             long long& cap) {
    const long long nU = eval_nslots_u(e, K, S);
    if (s < nU) {
        if (e.offU) {
            k = e.slotcol[s];
            st = (s - e.slotoff[k]) * SEG;
            const long long o = e.offU[k];
            u = e.Uf + o;
            cap = e.offU[k + 1] - o;
        } else {
            k = s / S;
            st = (s - k * S) * SEG;
            u = e.Uf + k * e.ldU;
            cap = e.ldU;
        }
    } else {
        k = K;
        st = (s - nU) * SEG;
        u = e.Vf;
        cap = e.capV;
    }
}

__device__ __forceinline__ void eval_prefetch(const EvalArgs& e, long long K, long long s,
                                              long long nslots, EvalPrefetch& pf) {
    const int lane = threadIdx.x & 63;
    pf.L = 0;
#pragma unroll
    for (int j = 0; j < SEG / 64; ++j) pf.u[j] = 0.0;
    if (s >= nslots) return;
    const long long S = slots_per_column(e.ldU);
    long long k, st, cap;
    const double* u;
    eval_slot_of(e, K, S, s, k, st, u, cap);
    long long L = k < K ? e.lenU[k] : e.lenV[0];
    pf.L = L > cap ? cap : L;
#pragma unroll
    for (int j = 0; j < SEG / 64; ++j) {
        const long long n = st + lane + 64 * j;
        if (n < cap) pf.u[j] = u[n];  // within the row's allocation whatever its length
    }
}

// Segments s = wave0, wave0 + nwaves, ... of the trial point: the three partial sums of each.
// pf (or NULL): the prefetched length and terms of segment wave0.
__device__ __forceinline__ void eval_segments(const Ctrl* ctrl, const EvalArgs& e, long long K,
                                              const double* alpha, Partial* part, long long wave0,
                                              long long nwaves, const EvalPrefetch* pf = nullptr) {
    if (ctrl->done || !ctrl->trial_valid) return;
    const int lane = threadIdx.x & 63;
    const long long S = slots_per_column(e.ldU);
    const long long nslots = eval_nslots_u(e, K, S) + slots_per_column(e.capV);
    for (long long s = wave0; s < nslots; s += nwaves) {
        long long k, st, cap;
        const double* u;
        eval_slot_of(e, K, S, s, k, st, u, cap);
        const bool pre = pf != nullptr && s == wave0;
        long long L = pre ? pf->L : (k < K ? e.lenU[k] : e.lenV[0]);
        if (L > cap) L = cap;
        if (st >= L) continue;  // wave-uniform: the column ends before this slot
        const double a = k < K ? alpha[k] : ctrl->A_trial;
        const long long end = st + SEG < L ? st + SEG : L;
        DmeAcc s0{0.0, 0.0}, s1{0.0, 0.0};
        double s2 = 0.0;
#pragma unroll
        for (int j = 0; j < SEG / 64; ++j) {
            const long long n = st + lane + 64 * j;
            if (n < end) {
                const double un = pre ? pf->u[j] : u[n];
                const double x = __dadd_rn(a, (double)n);
                s0.add(__dmul_rn(un, log(x)));                 // U[k][n] * log(alpha_k + n)   :21 / :25
                s1.add(__ddiv_rn(un, x));                      // U[k][n] / (alpha_k + n)      :42 / :37
                s2 = __dadd_rn(s2, __ddiv_rn(un, __dmul_rn(x, x)));  // U/(alpha_k+n)^2       :60 / :53
            }
        }
        s0 = dme_wave_sum_acc(s0);
        s1 = dme_wave_sum_acc(s1);
        s2 = dme_wave_sum_f64(s2);
        if (lane == 0) {
            Partial p{s0.hi, s0.lo, s1.hi, s1.lo, s2, 0.0};
            part[s] = p;
        }
    }
}

__global__ __launch_bounds__(EVAL_BLOCK) void eval_kernel(const Ctrl* ctrl, EvalArgs e, long long K,
                                                          const double* alpha, Partial* part) {
    eval_segments(ctrl, e, K, alpha, part, ((long long)blockIdx.x * EVAL_BLOCK + threadIdx.x) >> 6,
                  ((long long)gridDim.x * EVAL_BLOCK) >> 6);
}

// ---------------------------------------------------------------------------------------------
// shared device pieces of the step kernel
// ---------------------------------------------------------------------------------------------
struct Gathered {
    DmeAcc logsum;  // sum_k sum_n U log(.)     (block total)
    DmeAcc t0, t1;  // V sums
    double t2;
    // by-products at the evaluated point, so that an accepted point needs no further reductions:
    double gradSq;          // sum_k g_k^2                       (sqVectorSize, :104)
    double numSum, denSum;  // sum g_k/d_k, sum 1/d_k            (getPredictedStep, :69-70)
    bool zero_d;            // some d_k == 0 (ZeroDivisionError at :69)
    double sumAlphaBeta;    // sum alpha_k Beta_k      (CPT/dme.py:32-34), 0 without Beta
    double sumLgamma;       // sum lgamma(alpha_k)      (CPT/dme.py:38),    0 without W
};

// The partial sums one thread gathers, loaded before they are needed (iter_kernel: at kernel start,
// beside the state): the slots of its NCOL columns k = tid, tid + STEP_BLOCK, ... (at most PF_SMAX
// each) and, for thread t < SV, slot t of V.  Loaded whatever the columns' lengths turn out to be -
// the table holds every slot - and masked by the lengths afterwards.
constexpr int PF_SMAX = 2;  // prefetch covers ldU <= 2 * SEG = 512 count bins
struct PartV {
    double v[5];  // s0hi, s0lo, s1hi, s1lo, s2
};
__device__ __forceinline__ PartV ld_part(const Partial* p) {
    return PartV{{p->s0hi, p->s0lo, p->s1hi, p->s1lo, p->s2}};
}
// Adds the slots [0, cnt) of one column in index order (`first`: slot 0 already in registers, the loop
// then starts at `from` = 1).  Four slots are loaded before the first of them is added: the adds
// keep their order (same bits as one by one), but a long column - heavy-tailed counts: 38 slots at
// 9700 bins - no longer pays a memory round trip per slot (BASELINE configs[4]: 0.45 -> 0.36 ms
// for the 13-iteration solve).
// DEEP: eight slots in flight for the long columns of heavy-tailed counts (the kernels without the
// (Beta, W) terms that take their partial sums from the table; elsewhere the 40 registers it takes spill).
template <bool DEEP>
__device__ __forceinline__ void gather_slots(const Partial* base, long long from, long long cnt,
                                             const PartV* first, DmeAcc& a0, DmeAcc& a1, double& a2) {
    auto add = [&](const PartV& p) {
        a0.add(DmeAcc{p.v[0], p.v[1]});
        a1.add(DmeAcc{p.v[2], p.v[3]});
        a2 = __dadd_rn(a2, p.v[4]);
    };
    if (first) add(*first);
    long long i = from;
    if constexpr (DEEP) {
        for (; i + 7 < cnt; i += 8) {  // (38 slots are five round trips instead of ten)
            PartV p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = ld_part(base + i + u);
#pragma unroll
            for (int u = 0; u < 8; ++u) add(p[u]);
        }
    }
    for (; i + 3 < cnt; i += 4) {
        const PartV p0 = ld_part(base + i), p1 = ld_part(base + i + 1), p2 = ld_part(base + i + 2),
                    p3 = ld_part(base + i + 3);
        add(p0);
        add(p1);
        add(p2);
        add(p3);
    }
    for (; i < cnt; ++i) add(ld_part(base + i));
}

template <int NCOL>
struct GatherPrefetch {
    long long L[NCOL > 0 ? NCOL : 1];
    PartV p[NCOL > 0 ? NCOL : 1];  // slot 0 of each column (most columns end there); others on demand
    long long LV;
    PartV pv;
};

template <int NCOL>
__device__ __forceinline__ void gather_prefetch(const EvalArgs& e, const Partial* part, long long K,
                                                GatherPrefetch<NCOL>& pf) {
    const int tid = threadIdx.x;
    const long long S = slots_per_column(e.ldU), SV = slots_per_column(e.capV);
    long long lv = e.lenV[0];
    pf.LV = lv > e.capV ? e.capV : lv;
    pf.pv = PartV{{0, 0, 0, 0, 0}};
    // up to 64 slots: every wave holds them all, one per lane (gather_eval sums them inside the wave)
    const long long vslot = SV <= 64 ? (long long)(tid & 63) : (long long)tid;
    if (vslot < SV) pf.pv = ld_part(part + eval_nslots_u(e, K, S) + vslot);
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const long long k = tid + (long long)c * STEP_BLOCK;
        pf.L[c] = 0;
        pf.p[c] = PartV{{0, 0, 0, 0, 0}};
        if (k < K) {
            pf.L[c] = eval_col_len(e, k);
            pf.p[c] = ld_part(part + eval_slot_base(e, k, S));
        }
    }
}

#ifdef DME_SOLVE_TS
__device__ long long g_dbg[16];
#define DBG_TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#else
#define DBG_TS(i)
#endif

// Per-column sums at the evaluated point -> g_out[k], d_out[k]; returns the block totals.
// Two fused block reductions: the V sums first (the gradient needs them), then everything per k.
// CPT = false compiles the (Beta, W) terms out, so the scalar-hyperprior kernels keep their
// register budget (116 VGPRs against 256 + spills with digamma / trigamma / lgamma inlined).
// NCOL > 0: the partial sums come from `pf` (gather_prefetch; needs K <= NCOL * STEP_BLOCK,
// S <= PF_SMAX, SV <= STEP_BLOCK); NCOL == 0: from the table.  Same values, same order.
template <bool CPT, int NCOL>
__device__ Gathered gather_eval(const Ws& w, const EvalArgs& e, long long K, double A,
                                const Prior& pr, const double* alpha, double* g_out, double* d_out,
                                BlockShared& sh, const GatherPrefetch<NCOL>* pf) {
    Gathered G;
    const int tid = threadIdx.x;
    const long long S = slots_per_column(e.ldU);
    const long long nU = eval_nslots_u(e, K, S);
    // V = column K
    {
        long long LV;
        if constexpr (NCOL > 0) {
            LV = pf->LV;
        } else {
            LV = e.lenV[0];
            if (LV > e.capV) LV = e.capV;
        }
        const long long nv = LV > 0 ? (LV + SEG - 1) / SEG : 0;
        DmeAcc a[2] = {{0, 0}, {0, 0}};
        double d[1] = {0.0};
        if (slots_per_column(e.capV) <= 64) {
            // Up to 64 slots (row totals below 16 384: every BASELINE configuration but the heavy
            // tail): each wave adds them all by itself, a slot per lane, in the order of the wave
            // sum - no LDS, no barrier.  (The block reduction below gives the same bits: its other
            // waves add zeros.  It costs 1.3 us of the step's critical path, this 0.4.)
            const int lane = tid & 63;
            PartV p{{0, 0, 0, 0, 0}};
            if (lane < nv) {
                if constexpr (NCOL > 0) p = pf->pv;
                else p = ld_part(w.part + nU + lane);
            }
            if (nv <= 2) {
                // One or two slots (row totals below 512: BASELINE configs[1] and [2]).  In the wave sum
                // lane 0 first adds lane 1 and from then on only zeros, and adding a zero pair changes
                // nothing after the first time: (slot 0 + slot 1) + 0 gives the same bits without the
                // seven dependent stages of each of the three wave sums (0.64 -> 0.2 us of the step).
                PartV q0, q1;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    q0.v[i] = dme_readlane_f64(p.v[i], 0);
                    q1.v[i] = dme_readlane_f64(p.v[i], 1);
                }
                const DmeAcc z{0.0, 0.0};
                a[0] = DmeAcc{q0.v[0], q0.v[1]};
                a[0].add(DmeAcc{q1.v[0], q1.v[1]});
                a[0].add(z);
                a[1] = DmeAcc{q0.v[2], q0.v[3]};
                a[1].add(DmeAcc{q1.v[2], q1.v[3]});
                a[1].add(z);
                d[0] = __dadd_rn(__dadd_rn(q0.v[4], q1.v[4]), 0.0);
            } else {
                a[0] = dme_wave_sum_acc(DmeAcc{p.v[0], p.v[1]});
                a[1] = dme_wave_sum_acc(DmeAcc{p.v[2], p.v[3]});
                d[0] = dme_wave_sum_f64(p.v[4]);
            }
        } else {
            for (long long i = tid; i < nv; i += STEP_BLOCK) {
                PartV p;
                if constexpr (NCOL > 0) p = pf->pv;  // SV <= STEP_BLOCK: i == tid
                else p = ld_part(w.part + nU + i);
                a[0].add(DmeAcc{p.v[0], p.v[1]});
                a[1].add(DmeAcc{p.v[2], p.v[3]});
                d[0] = __dadd_rn(d[0], p.v[4]);
            }
            bool none = false;
            block_reduce_fused<2, 1>(a, d, none, sh);
        }
        G.t0 = a[0];
        G.t1 = a[1];
        G.t2 = d[0];
    }
    DBG_TS(8);
    double psiA = 0.0;
    if (CPT && pr.has_W) {
        G.t2 = __dadd_rn(G.t2, __dmul_rn(pr.W, dme_trigamma(A)));  // CPT/dme.py:74-75
        psiA = dme_digamma(A);
    }
    const double termToSubtract = G.t1.value();
    const double hs = __dmul_rn(pr.lambda, A);  // hyperprior * sum(priorList)   DE/dme.py:46
    DmeAcc ls[1] = {{0, 0}};
    constexpr int NR = CPT ? 5 : 3;
    double r[NR] = {};  // gradSq, numSum, denSum, sum a*Beta, sum lgamma(a)
    bool zero = false;
    auto column = [&](long long k, const DmeAcc& a0, const DmeAcc& a1, double a2) {
        ls[0].add(a0);
        double g = __dsub_rn(a1.value(), termToSubtract);  // :45
        g = __dadd_rn(g, hs);                              // :46
        double dk = -a2;                                   // :60
        if (CPT && pr.beta) {
            const double bk = pr.beta[k];
            g = __dsub_rn(g, bk);                                          // CPT/dme.py:60
            r[NR - 2] = __dadd_rn(r[NR - 2], __dmul_rn(alpha[k], bk));     // CPT/dme.py:34
        }
        if (CPT && pr.has_W) {
            const double ak = alpha[k];
            g = __dadd_rn(g, __dmul_rn(pr.W, __dsub_rn(psiA, dme_digamma(ak))));  // CPT/dme.py:64
            dk = __dsub_rn(dk, __dmul_rn(pr.W, dme_trigamma(ak)));                // CPT/dme.py:86
            r[NR - 1] = __dadd_rn(r[NR - 1], lgamma(ak));                         // CPT/dme.py:38
        }
        g_out[k] = g;
        d_out[k] = dk;
        r[0] = __dadd_rn(r[0], __dmul_rn(g, g));
        if (dk == 0.0) {
            zero = true;
        } else {
            r[1] = __dadd_rn(r[1], __ddiv_rn(g, dk));
            r[2] = __dadd_rn(r[2], __ddiv_rn(1.0, dk));
        }
    };
    if constexpr (NCOL > 0) {
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const long long k = tid + (long long)c * STEP_BLOCK;
            if (k < K) {
                const long long L = pf->L[c];
                const long long cnt = L > 0 ? (L + SEG - 1) / SEG : 0;
                DmeAcc a0{0, 0}, a1{0, 0};
                double a2 = 0.0;
                if (cnt > 0) gather_slots<false>(w.part + eval_slot_base(e, k, S), 1, cnt, &pf->p[c], a0, a1, a2);
                column(k, a0, a1, a2);
            }
        }
    } else {
        for (long long k = tid; k < K; k += STEP_BLOCK) {
            const long long L = eval_col_len(e, k);
            const long long cnt = L > 0 ? (L + SEG - 1) / SEG : 0;
            DmeAcc a0{0, 0}, a1{0, 0};
            double a2 = 0.0;
            gather_slots<!CPT>(w.part + eval_slot_base(e, k, S), 0, cnt, nullptr, a0, a1, a2);
            column(k, a0, a1, a2);
        }
    }
    DBG_TS(9);
    block_reduce_fused<1, NR>(ls, r, zero, sh);
    DBG_TS(10);
    G.logsum = ls[0];
    G.gradSq = r[0];
    G.numSum = r[1];
    G.denSum = r[2];
    G.zero_d = zero;
    G.sumAlphaBeta = CPT ? r[NR - 2] : 0.0;
    G.sumLgamma = CPT ? r[NR - 1] : 0.0;
    return G;
}

template <bool CPT>
__device__ __forceinline__ double loss_from(const Gathered& G, double A, const Prior& pr) {
    // total = sum U log - sum V log + sum(alpha)*lambda ; loss = -total     DE/dme.py:21-29, :84
    DmeAcc t = G.logsum;
    DmeAcc neg{-G.t0.hi, -G.t0.lo};
    t.add(neg);
    t.add(__dmul_rn(A, pr.lambda));
    if (CPT && !pr.loss_ignores_prior) {
        if (pr.beta) t.add(-G.sumAlphaBeta);                                         // CPT/dme.py:32-34
        if (pr.has_W) t.add(__dmul_rn(pr.W, __dsub_rn(lgamma(A), G.sumLgamma)));   // CPT/dme.py:36-38
    }
    return -t.value();
}

// getPredictedStep (DE/dme.py:67-72) applied to alpha: trial = alpha + step.
// Returns false (zero division) where Python raises.
__device__ bool newton_trial(const double* alpha, const double* g, const double* d, double hconst,
                             long long K, double* trial, double* step_out, BlockShared& sh) {
    const int tid = threadIdx.x;
    bool zero = false;
    double num = 0.0, den = 0.0;
    for (long long k = tid; k < K; k += STEP_BLOCK) {
        const double dk = d[k];
        if (dk == 0.0) { zero = true; continue; }
        num = __dadd_rn(num, __ddiv_rn(g[k], dk));
        den = __dadd_rn(den, __ddiv_rn(1.0, dk));
    }
    if (block_any(zero, sh) || hconst == 0.0) return false;
    const double numSum = block_sum(num, sh), denSum = block_sum(den, sh);
    const double dd = __dadd_rn(__ddiv_rn(1.0, hconst), denSum);
    if (dd == 0.0) return false;
    const double b = __ddiv_rn(numSum, dd);
    for (long long k = tid; k < K; k += STEP_BLOCK) {
        const double st = __ddiv_rn(__dsub_rn(b, g[k]), d[k]);
        if (step_out) step_out[k] = st;
        if (trial) trial[k] = __dadd_rn(alpha[k], st);
    }
    return true;
}

// getPredictedStepAlt (DE/dme.py:76-80); trial = alpha * exp(step) with the bare-except of :168-171.
__device__ bool logspace_trial(const double* alpha, const double* g, const double* d, double hconst,
                               long long K, double* trial, double* step_out, BlockShared& sh) {
    const int tid = threadIdx.x;
    bool zero = false;
    double zs = 0.0, ss = 0.0;
    for (long long k = tid; k < K; k += STEP_BLOCK) {
        const double x = __dadd_rn(g[k], __dmul_rn(alpha[k], d[k]));  // :77
        if (x == 0.0) { zero = true; continue; }
        zs = __dadd_rn(zs, __ddiv_rn(alpha[k], x));
        ss = __dadd_rn(ss, __ddiv_rn(__dmul_rn(alpha[k], g[k]), x));
    }
    if (block_any(zero, sh) || hconst == 0.0) return false;
    const double Z = __dadd_rn(__ddiv_rn(1.0, hconst), block_sum(zs, sh));  // :78
    const double S = block_sum(ss, sh);                                     // :79
    if (Z == 0.0) return false;
    const double soz = __ddiv_rn(S, Z);
    for (long long k = tid; k < K; k += STEP_BLOCK) {
        const double x = __dadd_rn(g[k], __dmul_rn(alpha[k], d[k]));
        const double st = __ddiv_rn(__dsub_rn(soz, g[k]), x);  // :80
        if (step_out) step_out[k] = st;
        if (trial) {
            const double e = exp(st);
            // math.exp raises OverflowError on overflow; the bare except keeps the prior (:170-171)
            trial[k] = (isinf(e) && !isinf(st)) ? alpha[k] : __dmul_rn(alpha[k], e);
        }
    }
    return true;
}

// sum(trial) and the positivity test of testTrialPriors (DE/dme.py:98-100); returns whether the
// trial is valid (every alpha > 0) to every thread
__device__ bool finish_trial(Ctrl* ctrl, const double* trial, long long K, BlockShared& sh) {
    const int tid = threadIdx.x;
    double a[1] = {0.0};
    bool bad = false;
    for (long long k = tid; k < K; k += STEP_BLOCK) {
        const double t = trial[k];
        a[0] = __dadd_rn(a[0], t);
        if (t <= 0.0) bad = true;
    }
    block_reduce_plain<1>(a, bad, sh);
    if (tid == 0) {
        ctrl->A_trial = a[0];
        ctrl->trial_valid = bad ? 0 : 1;
    }
    return !bad;
}

// The start state of the loop, one launch.
// init_acc != NULL: the start point is the CLI's (DE/fdp.py:104-108) computed here from the mean
// accumulators, with dme_cli_init_alphas' own arithmetic (same bits), instead of being read from
// `alpha`.  status != NULL: the statistics pass's status words ride along in result[8..16), so one
// device-to-host copy at the end brings the solver's verdict and the data-error counters.
__global__ __launch_bounds__(STEP_BLOCK) void setup_kernel(Ws w, const double* alpha, long long K,
                                                           long long phase, const long long* lenU,
                                                           const long long* lenV, long long ldU,
                                                           long long capV, long long* result,
                                                           const double* init_acc, const u64* status) {
    static_assert(STEP_BLOCK == 512, "dme_cli_init_alphas sums with 512 threads too");
    __shared__ BlockShared sh;
    const int tid = threadIdx.x;
    if (result && tid < 8) result[tid] = 0;  // [6] is the done word the host polls
    if (result && status && tid < 8) result[8 + tid] = (long long)status[tid];
    if (tid == 0) {
        Ctrl* c = w.ctrl;
        c->nseg = w.nseg_max;
        c->done = 0; c->phase = phase; c->count = 0; c->n_evals = 0; c->exit_reason = 0;
        c->error = 0; c->launches = 0; c->branch = 0; c->currentLoss = 0.0; c->learnRate = 1.0;
        c->hconst_cur = 0.0; c->fragile = 0;
    }
    if (init_acc) {
        const double priorSum = dme_cli_prior_sum<STEP_BLOCK>(init_acc, K, sh.dbl);
        for (long long k = tid; k < K; k += STEP_BLOCK)
            w.alpha_trial[k] = __dadd_rn(__ddiv_rn(init_acc[k], priorSum), 0.01);
    } else {
        for (long long k = tid; k < K; k += STEP_BLOCK) w.alpha_trial[k] = alpha[k];
    }
    __syncthreads();
    finish_trial(w.ctrl, w.alpha_trial, K, sh);
}

// One round of the state machine of findDirichletPriors: consumes the evaluation of the trial
// point, decides, and leaves the next trial point (or the result).  Called by every thread of a
// STEP_BLOCK workgroup.
// iter_kernel keeps the state in LDS and loads the CURRENT point's vectors only when a launch needs
// them: an accepted trial overwrites them, so the common launch (Newton step accepted) never does.
struct LazyCur {
    const double* alpha_cur;
    const double* g_cur;
    const double* d_cur;
};

template <bool CPT, int NCOL = 0>
__device__ void step_body(const Ws& w, const EvalArgs& e, long long K, long long iterations,
                          const Prior& pr, double* trace, double* trace_alpha, long long trace_cap,
                          double* alpha_out, long long* result, double* final_loss,
                          BlockShared& sh, const bool writer = true,
                          const GatherPrefetch<NCOL>* pf = nullptr, const LazyCur* lz = nullptr) {
    // `writer`: this workgroup publishes the outputs (alpha_out, result, final_loss, the trace).  In
    // iter_kernel every workgroup runs this function redundantly on its own LDS copy of the state and
    // only workgroup 0 writes.
    Ctrl* c = w.ctrl;
    if (c->done) return;
    const int tid = threadIdx.x;
    const long long phase = c->phase;
    const bool valid = c->trial_valid != 0;
    const double A = c->A_trial;
    double currentLoss = c->currentLoss;
    double learnRate = c->learnRate;
    long long count = c->count;
    long long n_evals = c->n_evals;
    long long branch = c->branch;
    long long fragile = c->fragile;

    double lossT = INFINITY;
    double hconstT = 0.0;
    Gathered G{};
    DBG_TS(0);
    if (valid) {
        G = gather_eval<CPT, NCOL>(w, e, K, A, pr, w.alpha_trial, w.g_trial, w.d_trial, sh, pf);
        DBG_TS(1);
        lossT = loss_from<CPT>(G, A, pr);
        DBG_TS(2);
        hconstT = G.t2;
        n_evals += 1;
        // SURVEY 7 "parity-fragile": the comparisons of DE/dme.py:160 / :176 are about to be taken on
        // a loss difference of a few ulp, i.e. on rounding noise that a different summation order
        // (the reference's sequential sums) may resolve the other way
        if (phase != PH_INIT &&
            fabs(__dsub_rn(lossT, currentLoss)) <= 4.0 * 2.220446049250313e-16 * fabs(currentLoss))
            fragile += 1;
    }

    auto finish = [&](long long reason, long long err) {
        // returned priors are alpha_cur
        if (writer)
            for (long long k = tid; k < K; k += STEP_BLOCK) alpha_out[k] = w.alpha_cur[k];
        if (tid == 0) {
            c->done = 1; c->exit_reason = reason; c->error = err; c->count = count;
            c->n_evals = n_evals; c->currentLoss = currentLoss;
        }
        if (tid == 0 && writer) {
            result[0] = reason; result[1] = count; result[2] = n_evals; result[3] = err;
            result[6] = 1;  // done: what the host polls
            result[4] = c->launches + 1;
            result[5] = __double_as_longlong(currentLoss);
            result[7] = fragile;
            if (final_loss) final_loss[0] = currentLoss;
        }
    };
    auto store_state = [&](long long new_phase) {
        if (tid == 0) {
            c->phase = new_phase; c->count = count; c->n_evals = n_evals;
            c->currentLoss = currentLoss; c->learnRate = learnRate; c->branch = branch;
            c->fragile = fragile;
            c->launches += 1;
        }
    };

    // (a thread only ever touches its own columns k = tid, tid + STEP_BLOCK, ... of these vectors)
    auto need_cur = [&]() {
        if (lz)
            for (long long k = tid; k < K; k += STEP_BLOCK) {
                w.alpha_cur[k] = lz->alpha_cur[k];
                w.g_cur[k] = lz->g_cur[k];
                w.d_cur[k] = lz->d_cur[k];
            }
    };
    bool accept = false;
    if (phase == PH_INIT) {
        if (!valid) {  // math.log(x <= 0) -> ValueError at DE/dme.py:131
            for (long long k = tid; k < K; k += STEP_BLOCK) w.alpha_cur[k] = w.alpha_trial[k];
            finish(0, DME_ERR_MATH_DOMAIN);
            return;
        }
        currentLoss = lossT;  // :131
        accept = true;
        branch = 0;
    } else if (phase == PH_NEWTON) {
        if (lossT < currentLoss) {  // :160 (strict)
            currentLoss = lossT;
            accept = true;
            branch = 1;
        } else {
            // :165-172 log-space trial from the CURRENT point's gradient / Hessian
            need_cur();
            const bool ok = logspace_trial(w.alpha_cur, w.g_cur, w.d_cur, c->hconst_cur, K,
                                           w.alpha_trial, nullptr, sh);
            if (!ok) { finish(0, DME_ERR_ZERO_DIVISION); return; }
            __syncthreads();
            finish_trial(c, w.alpha_trial, K, sh);
            learnRate = 1.0;  // :175
            store_state(PH_LOGSPACE);
            return;
        }
    } else {  // PH_LOGSPACE / PH_HALVING: the while loop of :176-180
        need_cur();
        if (lossT > currentLoss) {
            // A trial with an alpha <= 0 scores inf without an evaluation (:98-100) and the loop
            // goes on: those rounds are walked here instead of costing a launch pair each (a bad
            // start on a large vocabulary takes ~200 of them).  The walk ends: the step shrinks
            // towards the current point, whose alphas are positive.
            // Rounds whose step is still far too long need no pass over the columns: with
            // r = min_k alpha_k / -g_k over the falling columns, learnRate > r (1 + 1e-9) leaves
            // that column at or below zero whatever the rounding.  The rate itself is still
            // multiplied round by round, so the sequence of rates is the reference's.
            double r = INFINITY;
            for (long long k = tid; k < K; k += STEP_BLOCK) {
                const double gk = w.g_cur[k];
                if (gk < 0.0) r = fmin(r, __ddiv_rn(w.alpha_cur[k], -gk));
            }
            r = __dmul_rn(block_min_f64(r, sh), 1.000000001);
            // (a rate that no longer shrinks - the smallest denormal times 0.9 rounds back to itself -
            // ends both loops: every wave must leave this kernel)
            while (__dmul_rn(learnRate, 0.9) > r && __dmul_rn(learnRate, 0.9) < learnRate)
                learnRate = __dmul_rn(learnRate, 0.9);
            bool ok;
            bool stuck = false;
            do {
                const double next = __dmul_rn(learnRate, 0.9);
                stuck = !(next < learnRate);
                learnRate = next;
                for (long long k = tid; k < K; k += STEP_BLOCK)
                    w.alpha_trial[k] = __dadd_rn(w.alpha_cur[k], __dmul_rn(w.g_cur[k], learnRate));
                __syncthreads();
                ok = finish_trial(c, w.alpha_trial, K, sh);
            } while (!ok && !stuck);
            if (!ok) {  // the reference would loop forever here (an infinite gradient component)
                finish(0, DME_ERR_INTERNAL);
                return;
            }
            store_state(PH_HALVING);
            return;
        }
        if (learnRate < 9.5367431640625e-07) {  // 2 ** -20, :134, :182-184
            finish(2, 0);
            return;
        }
        currentLoss = lossT;  // :186-187
        accept = true;
        branch = (phase == PH_LOGSPACE) ? 2 : 3;
    }

    if (accept) {
        for (long long k = tid; k < K; k += STEP_BLOCK) {
            w.alpha_cur[k] = w.alpha_trial[k];
            w.g_cur[k] = w.g_trial[k];
            w.d_cur[k] = w.d_trial[k];
        }
        if (tid == 0) c->hconst_cur = hconstT;
    }
    const double hconst = hconstT;  // == hconst_cur after an accept
    DBG_TS(3);

    // ---- top of the while loop, DE/dme.py:137-150 ---------------------------------------------
    if (count >= iterations) {  // :137 / :189-190
        finish(3, 0);
        return;
    }
    count += 1;
    const double gradSq = G.gradSq;  // :142 — an accepted point is always an evaluated one
    if (count - 1 < trace_cap && writer) {
        if (trace && tid == 0) {
            double* t = trace + (count - 1) * 4;
            t[0] = currentLoss; t[1] = gradSq; t[2] = (double)branch; t[3] = learnRate;
        }
        if (trace_alpha)
            for (long long k = tid; k < K; k += STEP_BLOCK)
                trace_alpha[(count - 1) * K + k] = w.alpha_cur[k];
    }
    if (gradSq < 0.0009765625) {  // 2 ** -10, :133, :145-147
        finish(1, 0);
        return;
    }
    DBG_TS(4);
    // getPredictedStep (:67-72) from the sums gathered with the evaluation
    if (G.zero_d || hconst == 0.0) { finish(0, DME_ERR_ZERO_DIVISION); return; }
    const double dd = __dadd_rn(__ddiv_rn(1.0, hconst), G.denSum);
    if (dd == 0.0) { finish(0, DME_ERR_ZERO_DIVISION); return; }
    const double bb = __ddiv_rn(G.numSum, dd);
    for (long long k = tid; k < K; k += STEP_BLOCK)
        w.alpha_trial[k] = __dadd_rn(w.alpha_cur[k], __ddiv_rn(__dsub_rn(bb, w.g_cur[k]), w.d_cur[k]));
    DBG_TS(5);
    finish_trial(c, w.alpha_trial, K, sh);
    DBG_TS(6);
    store_state(PH_NEWTON);
}

template <bool CPT>
__global__ __launch_bounds__(STEP_BLOCK) void step_kernel(Ws w, EvalArgs e, long long K,
                                                          long long iterations, Prior pr,
                                                          double* trace, double* trace_alpha,
                                                          long long trace_cap, double* alpha_out,
                                                          long long* result, double* final_loss) {
    __shared__ BlockShared sh;
    step_body<CPT>(w, e, K, iterations, pr, trace, trace_alpha, trace_cap, alpha_out, result,
                   final_loss, sh);
}

// Small problems (a few dozen segments: the reference's sample files, K = 2 Beta-binomial data):
// the whole loop in ONE launch.  The workgroup's eight waves evaluate the segments, a barrier,
// the same step_body, a barrier, and again - the arithmetic and its order are those of the
// eval_kernel / step_kernel chain, whose ~10 us per round was launch overhead on such inputs.
template <bool CPT>
__global__ __launch_bounds__(STEP_BLOCK) void small_solve_kernel(Ws w, EvalArgs e, long long K,
                                                                 long long iterations, Prior pr,
                                                                 double* trace, double* trace_alpha,
                                                                 long long trace_cap,
                                                                 double* alpha_out, long long* result,
                                                                 double* final_loss,
                                                                 long long max_rounds) {
    __shared__ BlockShared sh;
    const long long wave = threadIdx.x >> 6;
    for (long long round = 0;; ++round) {
        eval_segments(w.ctrl, e, K, w.alpha_trial, w.part, wave, STEP_WAVES);
        __syncthreads();
        step_body<CPT>(w, e, K, iterations, pr, trace, trace_alpha, trace_cap, alpha_out, result,
                       final_loss, sh);
        __syncthreads();
        if (((volatile Ctrl*)w.ctrl)->done) break;
        if (round >= max_rounds) {  // cannot happen for finite inputs; every wave must leave
            if (threadIdx.x == 0) {
                w.ctrl->done = 1;
                result[3] = DME_ERR_INTERNAL;
                result[6] = 1;
            }
            break;
        }
    }
}

// Mid-size problems (K up to FUSED_KMAX columns, more segments than one workgroup evaluates well):
// ONE launch per evaluation instead of an (eval, step) pair, and no set-up launch.  Launch j first
// consumes the partial sums of evaluation j - 1: EVERY workgroup runs the same step_body redundantly
// on its own LDS copy of the state (control block + the six K-vectors), so every workgroup knows the
// next trial point without any hand-off inside the launch; then the workgroups share out the
// segments of that point's evaluation.  Only workgroup 0 writes the state and the outputs, into the
// copy the next launch reads (two copies, alternating: no workgroup reads what another one writes
// in the same launch).  Launch 0 builds the start state instead of a step (from `alpha`, or from
// the CLI's mean accumulators).  Everything a launch needs from memory that does not depend on its
// own arithmetic - the state, the partial sums its threads gather (NCOL > 0), the terms of each
// wave's first segment - is loaded in ONE round at kernel start: a launch's critical path is one
// memory round trip, the step's block reductions and one segment's arithmetic.  Same device
// functions, same order of operations as the pair chain: the same bits
// (test_single_launch_and_chained_solvers_agree_bit_for_bit).  profiles/r2_notes.md has the times.
constexpr long long FUSED_KMAX = 2048;  // 6 K-vectors in LDS: 96 KiB

struct IterArgs {
    long long K, iterations;
    double* trace;
    double* trace_alpha;
    long long trace_cap;
    double* alpha;             // in (launch 0, when init_acc is NULL) / out
    long long* result;
    double* final_loss;
    const double* init_acc;    // launch 0: CLI start point from the mean accumulators, or NULL
    const u64* status;         // launch 0: copied to result[8..16), or NULL
    int parity, first;
#ifdef DME_SOLVE_TS
    int launch;
#endif
};

template <bool CPT, int NCOL>
__global__ __launch_bounds__(STEP_BLOCK) void iter_kernel(Ws w, EvalArgs e, Prior pr, IterArgs a) {
    extern __shared__ __attribute__((aligned(16))) double kvec[];  // [6][K]
    __shared__ BlockShared sh;
    __shared__ Ctrl lc;
    const int tid = threadIdx.x;
    const long long K = a.K;
    const int parity = a.parity;
    const bool first = a.first != 0;
    const bool writer = blockIdx.x == 0;
#ifdef DME_SOLVE_TS
    long long ts[6];
    ts[0] = wall_clock64();
#define TS(i) ts[i] = wall_clock64()
#else
#define TS(i)
#endif
    const Ctrl* cin = parity ? w.ctrl2 : w.ctrl;
    Ctrl* cout = parity ? w.ctrl : w.ctrl2;
    const Partial* part_in = parity ? w.part2 : w.part;
    Partial* part_out = parity ? w.part : w.part2;
    double* l_acur = kvec;
    double* l_gcur = kvec + K;
    double* l_dcur = kvec + 2 * K;
    double* l_atr = kvec + 3 * K;
    const long long wave0 = (long long)blockIdx.x * STEP_WAVES + (tid >> 6);
    const long long nslots = eval_nslots_u(e, K, slots_per_column(e.ldU)) + slots_per_column(e.capV);

    // ---- one round of loads: this wave's first segment, this thread's partial sums, the state -----
    EvalPrefetch epf;
    eval_prefetch(e, K, wave0, nslots, epf);
    GatherPrefetch<NCOL> gpf;
    if constexpr (NCOL > 0)
        if (!first) gather_prefetch<NCOL>(e, part_in, K, gpf);
    if (first) {
        // the start state (what setup_kernel builds for the other launch structures)
        if (tid == 0) {
            Ctrl z{};
            z.phase = PH_INIT;
            z.learnRate = 1.0;
            z.nseg = nslots;
            lc = z;
        }
        if (a.init_acc) {
            const double priorSum = dme_cli_prior_sum<STEP_BLOCK>(a.init_acc, K, sh.dbl);
            for (long long k = tid; k < K; k += STEP_BLOCK)
                l_atr[k] = __dadd_rn(__ddiv_rn(a.init_acc[k], priorSum), 0.01);
        } else {
            for (long long k = tid; k < K; k += STEP_BLOCK) l_atr[k] = a.alpha[k];
        }
        __syncthreads();
        finish_trial(&lc, l_atr, K, sh);
        if (writer && tid < 8) {
            a.result[tid] = 0;  // [6] is the done word the host polls
            if (a.status) a.result[8 + tid] = (long long)a.status[tid];
        }
    } else {
        const double* in_acur = parity ? w.alpha_cur2 : w.alpha_cur;
        const double* in_gcur = parity ? w.g_cur2 : w.g_cur;
        const double* in_dcur = parity ? w.d_cur2 : w.d_cur;
        const double* in_atr = parity ? w.alpha_trial2 : w.alpha_trial;
        if (tid < (int)(sizeof(Ctrl) / 8)) ((long long*)&lc)[tid] = ((const long long*)cin)[tid];
        for (long long k = tid; k < K; k += STEP_BLOCK) l_atr[k] = in_atr[k];
        const LazyCur lz{in_acur, in_gcur, in_dcur};
        __syncthreads();
        TS(1);
        if (lc.done) {  // uniform over the grid: the solve ended in an earlier launch
            if (writer && tid == 0) cout->done = 1;  // the launch after this one reads `cout`
            return;
        }
        Ws lw = w;
        lw.ctrl = &lc;
        lw.alpha_cur = l_acur;
        lw.g_cur = l_gcur;
        lw.d_cur = l_dcur;
        lw.alpha_trial = l_atr;
        lw.g_trial = kvec + 4 * K;
        lw.d_trial = kvec + 5 * K;
        lw.part = const_cast<Partial*>(part_in);
        step_body<CPT, NCOL>(lw, e, K, a.iterations, pr, a.trace, a.trace_alpha, a.trace_cap, a.alpha,
                             a.result, a.final_loss, sh, writer, NCOL > 0 ? &gpf : nullptr, &lz);
    }
    __syncthreads();
    TS(2);
    if (writer) {
        if (tid < (int)(sizeof(Ctrl) / 8)) ((long long*)cout)[tid] = ((const long long*)&lc)[tid];
        double* o_acur = parity ? w.alpha_cur : w.alpha_cur2;
        double* o_gcur = parity ? w.g_cur : w.g_cur2;
        double* o_dcur = parity ? w.d_cur : w.d_cur2;
        double* o_atr = parity ? w.alpha_trial : w.alpha_trial2;
        for (long long k = tid; k < K; k += STEP_BLOCK) {
            o_atr[k] = l_atr[k];
            if (!first) {
                o_acur[k] = l_acur[k];
                o_gcur[k] = l_gcur[k];
                o_dcur[k] = l_dcur[k];
            }
        }
    }
    // the evaluation of the new trial point (nothing to do once done, or for a trial with an
    // alpha <= 0, which scores inf without an evaluation: eval_segments checks both)
    TS(3);
    eval_segments(&lc, e, K, l_atr, part_out, wave0, (long long)gridDim.x * STEP_WAVES, &epf);
    TS(4);
#ifdef DME_SOLVE_TS
    if (blockIdx.x == 0 && tid == 0 && a.launch < 20 && a.trace_cap >= 45) {
        double* o = a.trace + (long long)(22 + a.launch) * 4;
        o[0] = (double)(ts[1] - ts[0]);
        o[1] = (double)(ts[2] - ts[1]);
        o[2] = (double)(ts[3] - ts[2]);
        o[3] = (double)(ts[4] - ts[3]);
        if (a.launch == 5) {
            double* q = a.trace + 40 * 4;
            for (int i = 0; i < 6; ++i) q[i] = (double)(g_dbg[i + 1] - g_dbg[i]);
            q[6] = (double)(g_dbg[8] - g_dbg[0]);
            q[7] = (double)(g_dbg[9] - g_dbg[8]);
            q[8] = (double)(g_dbg[10] - g_dbg[9]);
        }
    }
#endif
}

// dme_eval epilogue: publish the evaluated quantities
template <bool CPT>
__global__ __launch_bounds__(STEP_BLOCK) void publish_eval_kernel(Ws w, EvalArgs e, long long K,
                                                                  Prior pr, double* out_scalars,
                                                                  double* out_grad,
                                                                  double* out_hdiag) {
    __shared__ BlockShared sh;
    Ctrl* c = w.ctrl;
    const int tid = threadIdx.x;
    const double A = c->A_trial;
    if (!c->trial_valid) {  // some alpha <= 0: math.log would raise; nothing was evaluated
        for (long long k = tid; k < K; k += STEP_BLOCK) {
            if (out_grad) out_grad[k] = NAN;
            if (out_hdiag) out_hdiag[k] = NAN;
        }
        if (tid == 0) { out_scalars[0] = NAN; out_scalars[1] = NAN; out_scalars[2] = A; out_scalars[3] = NAN; }
        return;
    }
    const Gathered G = gather_eval<CPT, 0>(w, e, K, A, pr, w.alpha_trial, w.g_trial, w.d_trial, sh,
                                           (const GatherPrefetch<0>*)nullptr);
    const double loss = loss_from<CPT>(G, A, pr);
    for (long long k = tid; k < K; k += STEP_BLOCK) {
        if (out_grad) out_grad[k] = w.g_trial[k];
        if (out_hdiag) out_hdiag[k] = w.d_trial[k];
    }
    const double gradSq = G.gradSq;
    if (tid == 0) {
        out_scalars[0] = -loss;
        out_scalars[1] = G.t2;
        out_scalars[2] = A;
        out_scalars[3] = gradSq;
    }
}

__global__ __launch_bounds__(STEP_BLOCK) void predicted_step_kernel(double hconst,
                                                                    const double* hdiag,
                                                                    const double* grad,
                                                                    const double* alphas,
                                                                    long long K, double* step,
                                                                    long long* status, int alt) {
    __shared__ BlockShared sh;
    const bool ok = alt ? logspace_trial(alphas, grad, hdiag, hconst, K, nullptr, step, sh)
                        : newton_trial(nullptr, grad, hdiag, hconst, K, nullptr, step, sh);
    if (threadIdx.x == 0) status[0] = ok ? 0 : DME_ERR_ZERO_DIVISION;
}

// the slots of the K columns: dense-padded U gives every column those of the longest one
long long host_nslots_u(const EvalArgs& e, long long K) {
    return e.offU ? e.nslotsU : K * slots_per_column(e.ldU);
}

EvalArgs dense_args(const double* Uf, long long ldU, const long long* lenU, const double* Vf,
                    long long capV, const long long* lenV) {
    return EvalArgs{Uf, ldU, lenU, Vf, capV, lenV, nullptr, nullptr, nullptr, 0};
}

// dme_ragged_t -> EvalArgs (ldU: the longest column's allocation is not known here and not needed; the
// slot count stands in for it where a number identifies the shape)
int ragged_args(const double* Uf, const dme_ragged_t* rg, const long long* lenU, const double* Vf,
                long long capV, const long long* lenV, EvalArgs* out) {
    DME_REQUIRE(rg != nullptr && rg->offU && rg->slotoff && rg->slotcol, "null ragged layout");
    DME_REQUIRE(rg->nslots >= 0 && rg->nslots < (1LL << 31) && rg->total >= 0, "bad ragged layout");
    *out = EvalArgs{Uf, rg->nslots > 0 ? rg->nslots : 1, lenU, Vf, capV, lenV, (const long long*)rg->offU,
                    (const long long*)rg->slotoff, (const int*)rg->slotcol, (long long)rg->nslots};
    return DME_OK;
}

int check_common(const EvalArgs& e, long long K, void* workspace, long long workspace_bytes, Ws* w) {
    const long long ldU = e.ldU, capV = e.capV;
    DME_REQUIRE(K >= 1 && ldU >= 1 && capV >= 1, "bad shape K=%lld ldU=%lld capV=%lld", K, ldU, capV);
    DME_REQUIRE(K < (1LL << 31) && ldU < (1LL << 31) && capV < (1LL << 31), "shape too large");
    DME_REQUIRE(e.Uf && e.lenU && e.Vf && e.lenV, "null statistics pointer");
    DME_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0,
                "workspace must be a 256-byte aligned device pointer");
    *w = carve(workspace, K, host_nslots_u(e, K), capV);
    if (workspace_bytes < w->bytes)
        return dme_set_error(DME_ERR_WORKSPACE, "solver workspace: need %lld bytes, got %lld",
                             w->bytes, workspace_bytes);
    return DME_OK;
}

// 64 bytes of pinned host memory per host thread for the completion poll (never freed).
long long* poll_buffer() {
    static thread_local long long* buf = nullptr;
    if (!buf) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocDefault) != hipSuccess) return nullptr;
        buf = (long long*)p;
    }
    return buf;
}

unsigned eval_grid(long long nseg_max) {
    long long waves_per_block = EVAL_BLOCK / 64;
    long long g = (nseg_max + waves_per_block - 1) / waves_per_block;
    int cus = dme_device_cus();
    if (cus <= 0) cus = 256;
    const long long cap = (long long)cus * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int64_t dme_solver_workspace_bytes(int64_t K, int64_t ldU, int64_t capV) {
    if (K < 1 || ldU < 1 || capV < 1) return -1;
    return carve(nullptr, K, K * slots_per_column(ldU), capV).bytes;
}

extern "C" int64_t dme_solver_workspace_bytes_ragged(int64_t K, int64_t nslots, int64_t capV) {
    if (K < 1 || nslots < 0 || capV < 1) return -1;
    return carve(nullptr, K, nslots, capV).bytes;
}

namespace {

int eval_impl(const EvalArgs& ea, int64_t K, const double* alpha, Prior pr, double* out_scalars,
              double* out_grad, double* out_hdiag, void* workspace, int64_t workspace_bytes,
              void* stream) {
    Ws w;
    int rc = check_common(ea, K, workspace, workspace_bytes, &w);
    if (rc) return rc;
    const long long* lenU = ea.lenU;
    const long long* lenV = ea.lenV;
    const long long ldU = ea.ldU, capV = ea.capV;
    DME_REQUIRE(alpha && out_scalars, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(setup_kernel, dim3(1), dim3(STEP_BLOCK), 0, s, w, alpha, (long long)K,
                       (long long)PH_EVALONLY, (const long long*)lenU, (const long long*)lenV,
                       (long long)ldU, (long long)capV, (long long*)nullptr, (const double*)nullptr,
                       (const u64*)nullptr);
    // the evaluation itself ignores the positivity test (dirichLogProb has none); force it on
    hipLaunchKernelGGL(eval_kernel, dim3(eval_grid(w.nseg_max)), dim3(EVAL_BLOCK), 0, s,
                       (const Ctrl*)w.ctrl, ea, (long long)K, (const double*)w.alpha_trial, w.part);
    const bool cpt = pr.beta != nullptr || pr.has_W;
    if (cpt)
        hipLaunchKernelGGL(publish_eval_kernel<true>, dim3(1), dim3(STEP_BLOCK), 0, s, w, ea,
                           (long long)K, pr, out_scalars, out_grad, out_hdiag);
    else
        hipLaunchKernelGGL(publish_eval_kernel<false>, dim3(1), dim3(STEP_BLOCK), 0, s, w, ea,
                           (long long)K, pr, out_scalars, out_grad, out_hdiag);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

}  // namespace

extern "C" int dme_eval(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                        int64_t capV, const int64_t* lenV, int64_t K, const double* alpha,
                        double hyperprior, double* out_scalars, double* out_grad,
                        double* out_hdiag, void* workspace, int64_t workspace_bytes, void* stream) {
    const Prior pr{hyperprior, nullptr, 0.0, 0, 0};
    return eval_impl(dense_args(Uf, ldU, (const long long*)lenU, Vf, capV, (const long long*)lenV), K,
                     alpha, pr, out_scalars, out_grad, out_hdiag, workspace, workspace_bytes, stream);
}

extern "C" int dme_eval_cpt(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                            int64_t capV, const int64_t* lenV, int64_t K, const double* alpha,
                            const double* beta, double W, int has_W, double* out_scalars,
                            double* out_grad, double* out_hdiag, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    const Prior pr{0.0, beta, W, has_W ? 1 : 0, 0};
    return eval_impl(dense_args(Uf, ldU, (const long long*)lenU, Vf, capV, (const long long*)lenV), K,
                     alpha, pr, out_scalars, out_grad, out_hdiag, workspace, workspace_bytes, stream);
}

// dme_eval / dme_eval_cpt on the ragged layout (beta == NULL and has_W == 0: the scalar hyperprior).
extern "C" int dme_eval_ragged(const double* Uf, const dme_ragged_t* ragged, const int64_t* lenU,
                               const double* Vf, int64_t capV, const int64_t* lenV, int64_t K,
                               const double* alpha, double hyperprior, const double* beta, double W,
                               int has_W, double* out_scalars, double* out_grad, double* out_hdiag,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    EvalArgs ea;
    int rc = ragged_args(Uf, ragged, (const long long*)lenU, Vf, capV, (const long long*)lenV, &ea);
    if (rc) return rc;
    DME_REQUIRE(!(hyperprior != 0.0 && (beta || has_W)), "the scalar hyperprior and (Beta, W) do not combine");
    const Prior pr{hyperprior, beta, W, has_W ? 1 : 0, 0};
    return eval_impl(ea, K, alpha, pr, out_scalars, out_grad, out_hdiag, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// dme_eval_noalloc: dme_eval for callers that cannot allocate (HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc).  The solver workspace that dme_eval takes as an argument
// (control block, segment table, per-segment partial sums, six K-vectors) lives in a file-scope
// device array instead.  Every GPU holds its own copy of the array at its own address, so the
// address is looked up once per device with hipGetSymbolAddress (with that device current) and only
// that pointer - never the array's name, which is a host-side stand-in - is handed to kernels.
// Nothing in it has to start at zero: setup_kernel writes the whole control block, eval_kernel writes
// every partial that the gather later reads (slots beyond a column's length are never read).  Calls that overlap on one
// GPU (two streams, two threads) race on the array.
// ---------------------------------------------------------------------------------------------
constexpr long long kNoAllocScratchBytes = DME_NOALLOC_SCRATCH_BYTES;  // 8 MiB per GPU, held until the process ends
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
std::atomic<void*> g_scratch_ptr[kMaxDevices];  // zero-initialised
std::mutex g_scratch_mutex;

// Device address of `scratch` on the current device.  Looked up on the first call per device (under a
// mutex; hipGetSymbolAddress may load the code object there, which is not a capturable stream
// operation: that first call must be made eagerly, outside any HIP graph capture); every later call
// is one atomic load.
int noalloc_scratch(void** out) {
    int dev = 0;
    DME_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices)
        return dme_set_error(DME_ERR_UNSUPPORTED, "device ordinal %d beyond %d", dev, kMaxDevices);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (!p) {
            DME_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            if (!p || ((uintptr_t)p % 256) != 0)
                return dme_set_error(DME_ERR_HIP, "hipGetSymbolAddress(scratch) returned %p", p);
            g_scratch_ptr[dev].store(p, std::memory_order_release);
        }
    }
    *out = p;
    return DME_OK;
}

}  // namespace

extern "C" int dme_eval_noalloc(const double* Uf, int64_t ldU, const int64_t* lenU,
                                const double* Vf, int64_t capV, const int64_t* lenV, int64_t K,
                                const double* alpha, double hyperprior, double* out_scalars,
                                double* out_grad, double* out_hdiag, void* stream) {
    DME_REQUIRE(K >= 1 && ldU >= 1 && capV >= 1, "bad shape K=%lld ldU=%lld capV=%lld",
                (long long)K, (long long)ldU, (long long)capV);
    DME_REQUIRE(K < (1LL << 31) && ldU < (1LL << 31) && capV < (1LL << 31), "shape too large");
    const long long need = carve(nullptr, K, K * slots_per_column(ldU), capV).bytes;
    if (need > kNoAllocScratchBytes)
        return dme_set_error(DME_ERR_WORKSPACE,
                             "dme_eval_noalloc: K=%lld ldU=%lld capV=%lld needs %lld bytes of solver "
                             "workspace, the static scratch holds %lld; use dme_eval with a workspace",
                             (long long)K, (long long)ldU, (long long)capV, need,
                             kNoAllocScratchBytes);
    void* ws = nullptr;
    int rc = noalloc_scratch(&ws);
    if (rc) return rc;
    const Prior pr{hyperprior, nullptr, 0.0, 0, 0};
    return eval_impl(dense_args(Uf, ldU, (const long long*)lenU, Vf, capV, (const long long*)lenV), K,
                     alpha, pr, out_scalars, out_grad, out_hdiag, ws, kNoAllocScratchBytes, stream);
}

extern "C" int dme_predicted_step(double hconst, const double* hdiag, const double* grad, int64_t K,
                                  double* step, int64_t* status, void* stream) {
    DME_REQUIRE(K >= 1 && hdiag && grad && step && status, "bad arguments");
    hipLaunchKernelGGL(predicted_step_kernel, dim3(1), dim3(STEP_BLOCK), 0, (hipStream_t)stream,
                       hconst, hdiag, grad, (const double*)nullptr, (long long)K, step,
                       (long long*)status, 0);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

extern "C" int dme_predicted_step_alt(double hconst, const double* hdiag, const double* grad,
                                      const double* alphas, int64_t K, double* step,
                                      int64_t* status, void* stream) {
    DME_REQUIRE(K >= 1 && hdiag && grad && alphas && step && status, "bad arguments");
    hipLaunchKernelGGL(predicted_step_kernel, dim3(1), dim3(STEP_BLOCK), 0, (hipStream_t)stream,
                       hconst, hdiag, grad, alphas, (long long)K, step, (long long*)status, 1);
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

// Launch structure of dme_solve / dme_solve_cpt for the calling host thread (include/dme_hip.h).
static thread_local int g_solve_path = DME_SOLVE_PATH_AUTO;

extern "C" int dme_solve_set_path(int path) {
    if (path < DME_SOLVE_PATH_AUTO || path > DME_SOLVE_PATH_FUSED)
        return dme_set_error(DME_ERR_INVALID_ARG, "unknown solver path %d", path);
    g_solve_path = path;
    return DME_OK;
}

namespace {

// Everything a solve needs besides the workspace: the arguments of dme_solve / dme_solve_enqueue.
struct SolveCall {
    EvalArgs ea;
    long long K;
    double* alpha;
    long long iterations;
    Prior pr;
    const double* init_acc;  // start from the CLI's initial alphas (set-up computes them), or NULL
    const u64* status;       // statistics status words to carry into result[8..16), or NULL
    double* trace;
    double* trace_alpha;
    long long trace_cap;
    long long* result;
    double* final_loss;
};

int pick_path(long long K, long long nseg_max) {
    const int forced = g_solve_path;
    if (forced == DME_SOLVE_PATH_SINGLE || (forced == DME_SOLVE_PATH_AUTO && nseg_max <= SMALL_NSEG))
        return DME_SOLVE_PATH_SINGLE;
    if (K <= FUSED_KMAX && (forced == DME_SOLVE_PATH_FUSED || forced == DME_SOLVE_PATH_AUTO))
        return DME_SOLVE_PATH_FUSED;
    return DME_SOLVE_PATH_CHAIN;
}

// every reference iteration costs at most 2 + ~400 evaluations (0.9^n underflows the step)
long long max_launches(long long iterations) { return (iterations + 2) * 450 + 16; }

template <bool CPT, int NCOL>
int launch_iter(const Ws& w, const EvalArgs& ea, const Prior& pr, const IterArgs& a, unsigned grid,
                size_t lds, hipStream_t s) {
    auto kern = iter_kernel<CPT, NCOL>;
    static std::atomic<bool> attr_set[64];  // zero-initialised; racing first calls both set the attribute  // per instantiation and device
    int dev = 0;
    DME_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        DME_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(6 * FUSED_KMAX * 8)));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(STEP_BLOCK), lds, s, w, ea, pr, a);
    return DME_OK;
}

template <bool CPT>
int launch_iter_ncol(int ncol, const Ws& w, const EvalArgs& ea, const Prior& pr, const IterArgs& a,
                     unsigned grid, size_t lds, hipStream_t s) {
    switch (ncol) {
        case 1: return launch_iter<CPT, 1>(w, ea, pr, a, grid, lds, s);
        case 2: return launch_iter<CPT, 2>(w, ea, pr, a, grid, lds, s);
        case 4: return launch_iter<CPT, 4>(w, ea, pr, a, grid, lds, s);
        default: return launch_iter<CPT, 0>(w, ea, pr, a, grid, lds, s);
    }
}

// Enqueues launches [first, first + count) of a solve on `s`.  What one launch is depends on the
// structure: FUSED - one iter_kernel (launch 0 builds the start state and evaluates it, so n
// evaluations take n + 1 launches); CHAIN - an (eval, step) pair, launch 0 preceded by the set-up
// kernel; SINGLE - set-up and the whole loop (first == 0 only; count is ignored).  Nothing is
// waited for.
int enqueue_launches(const SolveCall& c, const Ws& w, int path, long long first, long long count,
                     hipStream_t s) {
    const bool cpt = c.pr.beta != nullptr || c.pr.has_W;
    const EvalArgs& ea = c.ea;
    if (first == 0 && path != DME_SOLVE_PATH_FUSED) {
        hipLaunchKernelGGL(setup_kernel, dim3(1), dim3(STEP_BLOCK), 0, s, w, (const double*)c.alpha,
                           c.K, (long long)PH_INIT, ea.lenU, ea.lenV, ea.ldU, ea.capV, c.result,
                           c.init_acc, c.status);
        DME_HIP_CHECK(hipGetLastError());
    }
    if (path == DME_SOLVE_PATH_SINGLE) {
        if (first != 0) return DME_OK;
        const long long rounds = max_launches(c.iterations);
        if (cpt)
            hipLaunchKernelGGL(small_solve_kernel<true>, dim3(1), dim3(STEP_BLOCK), 0, s, w, ea, c.K,
                               c.iterations, c.pr, c.trace, c.trace_alpha, c.trace_cap, c.alpha,
                               c.result, c.final_loss, rounds);
        else
            hipLaunchKernelGGL(small_solve_kernel<false>, dim3(1), dim3(STEP_BLOCK), 0, s, w, ea, c.K,
                               c.iterations, c.pr, c.trace, c.trace_alpha, c.trace_cap, c.alpha,
                               c.result, c.final_loss, rounds);
    } else if (path == DME_SOLVE_PATH_FUSED) {
        const size_t lds = (size_t)(6 * c.K * 8);
        // one wave per segment and round; every workgroup repeats the step, so no more of them than
        // the evaluation can use
        int cus = dme_device_cus();
        if (cus <= 0) cus = 256;
        long long fg = (w.nseg_max + STEP_WAVES - 1) / STEP_WAVES;
        if (fg > cus) fg = cus;
        if (fg < 1) fg = 1;
        // the gathered partial sums are prefetched when a thread's share is small and fixed
        int ncol = 0;
        if (!ea.offU && slots_per_column(ea.ldU) <= PF_SMAX && slots_per_column(ea.capV) <= STEP_BLOCK)
            ncol = c.K <= STEP_BLOCK ? 1 : (c.K <= 2 * STEP_BLOCK ? 2 : 4);
        for (long long j = first; j < first + count; ++j) {
            const IterArgs a{c.K, c.iterations, c.trace, c.trace_alpha, c.trace_cap, c.alpha, c.result,
                             c.final_loss, c.init_acc, c.status, (int)(j & 1), j == 0 ? 1 : 0
#ifdef DME_SOLVE_TS
                             , (int)j
#endif
            };
            const int rc = cpt ? launch_iter_ncol<true>(ncol, w, ea, c.pr, a, (unsigned)fg, lds, s)
                               : launch_iter_ncol<false>(ncol, w, ea, c.pr, a, (unsigned)fg, lds, s);
            if (rc) return rc;
        }
    } else {
        const unsigned grid = eval_grid(w.nseg_max);
        for (long long j = first; j < first + count; ++j) {
            hipLaunchKernelGGL(eval_kernel, dim3(grid), dim3(EVAL_BLOCK), 0, s, (const Ctrl*)w.ctrl, ea,
                               c.K, (const double*)w.alpha_trial, w.part);
            if (cpt)
                hipLaunchKernelGGL(step_kernel<true>, dim3(1), dim3(STEP_BLOCK), 0, s, w, ea, c.K,
                                   c.iterations, c.pr, c.trace, c.trace_alpha, c.trace_cap, c.alpha,
                                   c.result, c.final_loss);
            else
                hipLaunchKernelGGL(step_kernel<false>, dim3(1), dim3(STEP_BLOCK), 0, s, w, ea, c.K,
                                   c.iterations, c.pr, c.trace, c.trace_alpha, c.trace_cap, c.alpha,
                                   c.result, c.final_loss);
        }
    }
    DME_HIP_CHECK(hipGetLastError());
    return DME_OK;
}

// Solves come in series on similar statistics (EM rounds, re-estimation as rows arrive, the steps of a
// benchmark).  The first batch is enqueued while earlier work on the stream (the statistics pass)
// still runs, so its launches cost the host nothing, but a launch that finds the solve finished still
// costs ~5 us of device time and a batch that falls short a host round trip (~25 us): the first batch
// covers what this thread's previous solve of the same shape needed plus one spare, 16 evaluations
// otherwise; the spare is dropped once two solves in a row of that shape needed the same number (a
// series that has settled, such as re-estimation of the same statistics).
thread_local long long hint_K = -1, hint_ldU = -1, hint_launches = 0, hint_repeats = 0;

void remember_solve(long long K, long long ldU, long long launches) {
    hint_repeats = (hint_K == K && hint_ldU == ldU && hint_launches == launches) ? hint_repeats + 1 : 0;
    hint_K = K;
    hint_ldU = ldU;
    hint_launches = launches;
}

long long first_batch(long long K, long long ldU, int path) {
    long long evals = 16;
    if (hint_K == K && hint_ldU == ldU && hint_launches > 0) {
        evals = hint_launches + (hint_repeats >= 1 ? 0 : 1);
        if (evals < 4) evals = 4;
        if (evals > 32) evals = 32;
    }
    return path == DME_SOLVE_PATH_FUSED ? evals + 1 : evals;  // FUSED: the first launch only evaluates
}

int check_call(const SolveCall& c) {
    DME_REQUIRE(c.alpha && c.result, "null pointer");
    DME_REQUIRE(c.iterations >= 0 && c.trace_cap >= 0, "negative iterations / trace_cap");
    return DME_OK;
}

int result_error(const long long* res) {
    if (res[3] == DME_ERR_ZERO_DIVISION)
        return dme_set_error(DME_ERR_ZERO_DIVISION, "float division by zero (an all-zero column or empty V)");
    if (res[3] == DME_ERR_MATH_DOMAIN)
        return dme_set_error(DME_ERR_MATH_DOMAIN, "math domain error (non-positive initial alpha)");
    if (res[3] != 0) return dme_set_error(DME_ERR_INTERNAL, "solver error %lld", res[3]);
    return DME_OK;
}

int solve_impl(const SolveCall& c, int64_t* result_host, void* workspace, int64_t workspace_bytes,
               void* stream) {
    Ws w;
    int rc = check_common(c.ea, c.K, workspace, workspace_bytes, &w);
    if (rc) return rc;
    if ((rc = check_call(c))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int path = pick_path(c.K, w.nseg_max);
    const long long limit = max_launches(c.iterations);
    long long* pinned = poll_buffer();
    if (!pinned) return dme_set_error(DME_ERR_HIP, "hipHostMalloc of the poll buffer failed");
    long long launched = 0, batch = first_batch(c.K, c.ea.ldU, path), done = 0;
    while (!done) {
        if ((rc = enqueue_launches(c, w, path, launched, batch, s))) return rc;
        launched += batch;
        // one 64-byte copy brings the done flag and, once it is set, the whole result block
        DME_HIP_CHECK(hipMemcpyAsync(pinned, c.result, 8 * sizeof(long long), hipMemcpyDeviceToHost, s));
        DME_HIP_CHECK(hipStreamSynchronize(s));
        done = pinned[6];
        if (!done && (launched > limit || path == DME_SOLVE_PATH_SINGLE))
            return dme_set_error(DME_ERR_INTERNAL, "solver did not finish after %lld launches", launched);
        batch = launched < 64 ? 8 : 32;
    }
    remember_solve(c.K, c.ea.ldU, pinned[4]);  // evaluations' worth of launches that did work
    if (result_host)
        for (int i = 0; i < 8; ++i) result_host[i] = pinned[i];
    return result_error(pinned);
}

}  // namespace

extern "C" int dme_solve(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                         int64_t capV, const int64_t* lenV, int64_t K, double* alpha,
                         int64_t iterations, double hyperprior, double* trace, double* trace_alpha,
                         int64_t trace_cap, int64_t* result, int64_t* result_host,
                         double* final_loss, void* workspace, int64_t workspace_bytes,
                         void* stream) {
    const SolveCall c{dense_args(Uf, ldU, (const long long*)lenU, Vf, capV, (const long long*)lenV), K,
                      alpha, iterations, Prior{hyperprior, nullptr, 0.0, 0, 0}, nullptr, nullptr, trace,
                      trace_alpha, trace_cap, (long long*)result, final_loss};
    return solve_impl(c, result_host, workspace, workspace_bytes, stream);
}

extern "C" int dme_solve_cpt(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                             int64_t capV, const int64_t* lenV, int64_t K, double* alpha,
                             int64_t iterations, const double* beta, double W, int has_W,
                             double* trace, double* trace_alpha, int64_t trace_cap, int64_t* result,
                             int64_t* result_host, double* final_loss, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    // inside findDirichletPriors the loss carries no prior term (CPT/dme.py:109-110)
    const SolveCall c{dense_args(Uf, ldU, (const long long*)lenU, Vf, capV, (const long long*)lenV), K,
                      alpha, iterations, Prior{0.0, beta, W, has_W ? 1 : 0, 1}, nullptr, nullptr, trace,
                      trace_alpha, trace_cap, (long long*)result, final_loss};
    return solve_impl(c, result_host, workspace, workspace_bytes, stream);
}

// dme_solve / dme_solve_cpt on the ragged layout (beta == NULL and has_W == 0: the scalar hyperprior).
extern "C" int dme_solve_ragged(const double* Uf, const dme_ragged_t* ragged, const int64_t* lenU,
                                const double* Vf, int64_t capV, const int64_t* lenV, int64_t K,
                                double* alpha, int64_t iterations, double hyperprior,
                                const double* beta, double W, int has_W, double* trace,
                                double* trace_alpha, int64_t trace_cap, int64_t* result,
                                int64_t* result_host, double* final_loss, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    EvalArgs ea;
    int rc = ragged_args(Uf, ragged, (const long long*)lenU, Vf, capV, (const long long*)lenV, &ea);
    if (rc) return rc;
    const bool cpt = beta != nullptr || has_W;
    DME_REQUIRE(!(hyperprior != 0.0 && cpt), "the scalar hyperprior and (Beta, W) do not combine");
    const SolveCall c{ea, K, alpha, iterations, Prior{hyperprior, beta, W, has_W ? 1 : 0, cpt ? 1 : 0},
                      nullptr, nullptr, trace, trace_alpha, trace_cap, (long long*)result, final_loss};
    return solve_impl(c, result_host, workspace, workspace_bytes, stream);
}

namespace {

int enqueue_impl(const SolveCall& c, int64_t* result_host, void* workspace, int64_t workspace_bytes,
                 int64_t first, int64_t count, int64_t* enqueued, void* stream) {
    Ws w;
    int rc = check_common(c.ea, c.K, workspace, workspace_bytes, &w);
    if (rc) return rc;
    if ((rc = check_call(c))) return rc;
    DME_REQUIRE(first >= 0 && count >= 0 && result_host && enqueued, "bad first / count / null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int path = pick_path(c.K, w.nseg_max);
    if (first > max_launches(c.iterations))
        return dme_set_error(DME_ERR_INTERNAL, "solver did not finish after %lld launches", (long long)first);
    if (path == DME_SOLVE_PATH_SINGLE && first > 0) {
        // the single launch runs the whole loop and always sets `done`: asking for more launches means
        // its result was lost, and enqueueing nothing while reporting launches would loop the caller
        *enqueued = 0;
        return dme_set_error(DME_ERR_INTERNAL, "the single-launch solver takes no further launches");
    }
    if (count == 0) count = first == 0 ? first_batch(c.K, c.ea.ldU, path) : (first < 64 ? 8 : 32);
    if ((rc = enqueue_launches(c, w, path, first, count, s))) return rc;
    *enqueued = count;
    DME_HIP_CHECK(hipMemcpyAsync(result_host, c.result, 16 * sizeof(long long), hipMemcpyDeviceToHost, s));
    return DME_OK;
}

}  // namespace

extern "C" int dme_solve_enqueue(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                                 int64_t capV, const int64_t* lenV, int64_t K, double* alpha,
                                 int64_t iterations, double hyperprior, const double* init_acc,
                                 const uint64_t* status, double* trace, double* trace_alpha,
                                 int64_t trace_cap, int64_t* result, int64_t* result_host,
                                 void* workspace, int64_t workspace_bytes, int64_t first,
                                 int64_t count, int64_t* enqueued, void* stream) {
    const SolveCall c{dense_args(Uf, ldU, (const long long*)lenU, Vf, capV, (const long long*)lenV), K,
                      alpha, iterations, Prior{hyperprior, nullptr, 0.0, 0, 0}, init_acc,
                      (const u64*)status, trace, trace_alpha, trace_cap, (long long*)result, nullptr};
    return enqueue_impl(c, result_host, workspace, workspace_bytes, first, count, enqueued, stream);
}

// dme_solve_enqueue on the ragged layout; dme_solve_finish(result_host, K, ragged->nslots) afterwards.
extern "C" int dme_solve_enqueue_ragged(const double* Uf, const dme_ragged_t* ragged,
                                        const int64_t* lenU, const double* Vf, int64_t capV,
                                        const int64_t* lenV, int64_t K, double* alpha,
                                        int64_t iterations, double hyperprior, const double* init_acc,
                                        const uint64_t* status, double* trace, double* trace_alpha,
                                        int64_t trace_cap, int64_t* result, int64_t* result_host,
                                        void* workspace, int64_t workspace_bytes, int64_t first,
                                        int64_t count, int64_t* enqueued, void* stream) {
    EvalArgs ea;
    int rc = ragged_args(Uf, ragged, (const long long*)lenU, Vf, capV, (const long long*)lenV, &ea);
    if (rc) return rc;
    const SolveCall c{ea, K, alpha, iterations, Prior{hyperprior, nullptr, 0.0, 0, 0}, init_acc,
                      (const u64*)status, trace, trace_alpha, trace_cap, (long long*)result, nullptr};
    return enqueue_impl(c, result_host, workspace, workspace_bytes, first, count, enqueued, stream);
}

extern "C" int dme_solve_finish(const int64_t* result_host, int64_t K, int64_t ldU) {
    DME_REQUIRE(result_host != nullptr, "null pointer");
    if (!result_host[6]) return DME_OK;  // not done: the caller enqueues more launches
    remember_solve(K, ldU, result_host[4]);
    return result_error((const long long*)result_host);
}
