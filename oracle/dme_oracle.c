/*
 * dme_oracle.c — CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C, single-threaded, loop-faithful restatement of the reference's
 * Dirichlet-multinomial hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the shipped package (bayespy_amd/) never does.
 *
 * Reference: maxsklar/BayesPy, DirichletEstimation/dirichletMultinomialEstimation.py ("DE/dme.py")
 * and DirichletEstimation/findDirichletPrior.py ("DE/fdp.py").  Every function cites the lines it
 * follows.  Summation ORDER, the use of pow(x, 2.0) for `x ** 2`, libm log()/exp() (what CPython's
 * math.log / math.exp call for finite positive doubles) and the branch structure are kept exactly,
 * so on the same glibc this file is bit-identical to the Python reference; that is pinned by
 * tests/test_oracle_golden.py against fixtures produced by the reference itself
 * (tests/golden/make_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no fused multiply-add may be formed).
 *
 * Data layout: U is ragged, stored flat with offsets: U[k][n] = Uflat[off[k] + n],
 * len(U[k]) = off[k+1] - off[k].  appendRow builds int64 counts (Python ints in the reference);
 * the solver side takes U/V as doubles — exact for the integer case (the reference converts with
 * float(...) or int*float, same value), and what weighted rows produce anyway.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_ZERO_DIVISION (-1) /* Python ZeroDivisionError (DE/dme.py:69-71, 78-80)        */
#define ORACLE_CAPACITY (-2)      /* caller-provided U/V capacity too small                    */
#define ORACLE_MATH_DOMAIN (-3)   /* Python ValueError: math.log of a non-positive number     */

/* ------------------------------------------------------------------------------------------
 * a2: CompressedRowData.appendRow(row, weight=1)  — DE/dme.py:113-123, executed literally:
 *     for k: for j in range(row[k]): U[k][j] += 1      (:116-119)
 *     for j in range(sum(row)):      V[j]    += 1      (:121-123)
 * a3: the CLI's mean accumulator — DE/fdp.py:87-89:
 *     weightForMean = 1.0/(1.0+sum(row)); priors[i] += row[i]*weightForMean
 * U is held dense-padded here: U[k*capU + j]; lenU[k] = len(U[k]).  Negative counts follow
 * Python: range(negative) is empty, sum(row) still includes them.
 * ---------------------------------------------------------------------------------------- */
#define DEFINE_APPEND_ROWS(NAME, T)                                                             \
    int NAME(const T* X, int64_t N, int64_t K, int64_t ld, int64_t* U, int64_t capU,            \
             int64_t* lenU, int64_t* V, int64_t capV, int64_t* lenV, double* init_acc) {        \
        for (int64_t i = 0; i < N; ++i) {                                                       \
            const T* row = X + i * ld;                                                          \
            int64_t total = 0;                                                                  \
            for (int64_t k = 0; k < K; ++k) {                                                   \
                int64_t x = (int64_t)row[k];                                                    \
                total += x;                                                                     \
                if (x > capU) return ORACLE_CAPACITY;                                           \
                int64_t* Uk = U + k * capU;                                                     \
                for (int64_t j = 0; j < x; ++j) Uk[j] += 1;                                     \
                if (x > lenU[k]) lenU[k] = x;                                                   \
            }                                                                                   \
            if (total > capV) return ORACLE_CAPACITY;                                           \
            for (int64_t j = 0; j < total; ++j) V[j] += 1;                                      \
            if (total > *lenV) *lenV = total;                                                   \
            if (init_acc) {                                                                     \
                double weightForMean = 1.0 / (1.0 + (double)total);                             \
                for (int64_t k = 0; k < K; ++k) init_acc[k] += (double)row[k] * weightForMean;  \
            }                                                                                   \
        }                                                                                       \
        return ORACLE_OK;                                                                       \
    }

DEFINE_APPEND_ROWS(oracle_append_rows_i32, int32_t)
DEFINE_APPEND_ROWS(oracle_append_rows_u16, uint16_t)

/* CSR rows: same loops over the stored entries only (absent entries are zero counts and touch
 * nothing in DE/dme.py:116-119).  init_acc follows DE/fdp.py:87-89 for the stored entries; the
 * zero entries add 0*w = 0.0 there, which leaves a non-negative accumulator unchanged. */
int oracle_append_rows_csr_i32(const int64_t* rowptr, const int32_t* colidx, const int32_t* vals,
                               int64_t N, int64_t K, int64_t* U, int64_t capU, int64_t* lenU,
                               int64_t* V, int64_t capV, int64_t* lenV, double* init_acc) {
    for (int64_t i = 0; i < N; ++i) {
        int64_t total = 0;
        for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            int64_t k = colidx[p], x = vals[p];
            if (k < 0 || k >= K) return ORACLE_CAPACITY;
            total += x;
            if (x > capU) return ORACLE_CAPACITY;
            int64_t* Uk = U + k * capU;
            for (int64_t j = 0; j < x; ++j) Uk[j] += 1;
            if (x > lenU[k]) lenU[k] = x;
        }
        if (total > capV) return ORACLE_CAPACITY;
        for (int64_t j = 0; j < total; ++j) V[j] += 1;
        if (total > *lenV) *lenV = total;
        if (init_acc) {
            double weightForMean = 1.0 / (1.0 + (double)total);
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p)
                init_acc[colidx[p]] += (double)vals[p] * weightForMean;
        }
    }
    return ORACLE_OK;
}

/* appendRow(row, weight) with arbitrary weights (DE/dme.py:113-123 literally; U/V become floats as
 * soon as a fractional weight is added).  U dense-padded doubles U[k*capU + j]. */
int oracle_append_rows_weighted_i32(const int32_t* X, const double* weights, int64_t N, int64_t K,
                                    int64_t ld, double* U, int64_t capU, int64_t* lenU, double* V,
                                    int64_t capV, int64_t* lenV) {
    for (int64_t i = 0; i < N; ++i) {
        const int32_t* row = X + i * ld;
        const double w = weights[i];
        int64_t total = 0;
        for (int64_t k = 0; k < K; ++k) {
            int64_t x = row[k];
            total += x;
            if (x > capU) return ORACLE_CAPACITY;
            double* Uk = U + k * capU;
            for (int64_t j = 0; j < x; ++j) Uk[j] += w;
            if (x > lenU[k]) lenU[k] = x;
        }
        if (total > capV) return ORACLE_CAPACITY;
        for (int64_t j = 0; j < total; ++j) V[j] += w;
        if (total > *lenV) *lenV = total;
    }
    return ORACLE_OK;
}

/* a3 (second half): DE/fdp.py:104-108.  priorSum = sum(priors) + 0.01; p/=priorSum; p+=0.01 */
void oracle_cli_init_alphas(const double* init_acc, int64_t K, double* alpha0) {
    double s = 0.0;
    for (int64_t k = 0; k < K; ++k) s += init_acc[k]; /* Python sum(): left to right from 0 */
    double priorSum = s + 0.01;
    for (int64_t k = 0; k < K; ++k) {
        double p = init_acc[k] / priorSum;
        alpha0[k] = p + 0.01;
    }
}

static double py_sum(const double* a, int64_t K) { /* sum(list): sequential from int 0 */
    double s = 0.0;
    for (int64_t k = 0; k < K; ++k) s += a[k];
    return s;
}

/* a4: dirichLogProb — DE/dme.py:16-29.  Returns ORACLE_MATH_DOMAIN where math.log would raise. */
int oracle_logprob(const double* alpha, int64_t K, const double* Uflat, const int64_t* off,
                   const double* V, int64_t lenV, double hyperprior, double* out) {
    double total = 0.0;
    for (int64_t k = 0; k < K; ++k) {
        const double* Uk = Uflat + off[k];
        int64_t n = off[k + 1] - off[k];
        for (int64_t i = 0; i < n; ++i) {
            double x = alpha[k] + (double)i;
            if (x <= 0.0) return ORACLE_MATH_DOMAIN;
            total += Uk[i] * log(x); /* :21 */
        }
    }
    double sumPrior = py_sum(alpha, K); /* :23 */
    for (int64_t i = 0; i < lenV; ++i) {
        double x = sumPrior + (double)i;
        if (x <= 0.0) return ORACLE_MATH_DOMAIN;
        total -= V[i] * log(x); /* :25 */
    }
    total += py_sum(alpha, K) * hyperprior; /* :28 */
    *out = total;
    return ORACLE_OK;
}

/* getTotalLoss — DE/dme.py:83-84: -1 * dirichLogProb */
int oracle_total_loss(const double* alpha, int64_t K, const double* Uflat, const int64_t* off,
                      const double* V, int64_t lenV, double hyperprior, double* out) {
    double lp;
    int rc = oracle_logprob(alpha, K, Uflat, off, V, lenV, hyperprior, &lp);
    if (rc) return rc;
    *out = -1 * lp;
    return ORACLE_OK;
}

/* a5: priorGradient — DE/dme.py:32-48 (note :46 adds hyperprior * sum(priorList)). */
int oracle_gradient(const double* alpha, int64_t K, const double* Uflat, const int64_t* off,
                    const double* V, int64_t lenV, double hyperprior, double* grad) {
    double termToSubtract = 0.0;
    for (int64_t i = 0; i < lenV; ++i) {
        double den = py_sum(alpha, K) + (double)i; /* :37 re-evaluates sum() every term */
        if (den == 0.0) return ORACLE_ZERO_DIVISION;
        termToSubtract += V[i] / den;
    }
    for (int64_t j = 0; j < K; ++j) {
        const double* Uj = Uflat + off[j];
        int64_t n = off[j + 1] - off[j];
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double den = alpha[j] + (double)i;
            if (den == 0.0) return ORACLE_ZERO_DIVISION;
            r += Uj[i] / den; /* :42 */
        }
        grad[j] = r;
    }
    double hs = hyperprior * py_sum(alpha, K);
    for (int64_t k = 0; k < K; ++k) {
        grad[k] -= termToSubtract; /* :45 */
        grad[k] += hs;             /* :46 */
    }
    return ORACLE_OK;
}

/* a6: priorHessianConst — DE/dme.py:52-53.  `x ** 2` on a Python float is libm pow(x, 2.0). */
int oracle_hconst(const double* alpha, int64_t K, const double* V, int64_t lenV, double* out) {
    double s = 0.0;
    for (int64_t i = 0; i < lenV; ++i) {
        double den = pow(py_sum(alpha, K) + (double)i, 2.0);
        if (den == 0.0) return ORACLE_ZERO_DIVISION;
        s += V[i] / den;
    }
    *out = s;
    return ORACLE_OK;
}

/* a7: priorHessianDiag — DE/dme.py:55-61 */
int oracle_hdiag(const double* alpha, int64_t K, const double* Uflat, const int64_t* off,
                 double* hdiag) {
    for (int64_t k = 0; k < K; ++k) {
        const double* Uk = Uflat + off[k];
        int64_t n = off[k + 1] - off[k];
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double den = pow(alpha[k] + (double)i, 2.0);
            if (den == 0.0) return ORACLE_ZERO_DIVISION;
            r -= Uk[i] / den; /* :60 */
        }
        hdiag[k] = r;
    }
    return ORACLE_OK;
}

/* a8: getPredictedStep — DE/dme.py:67-72 */
int oracle_predicted_step(double hConst, const double* hDiag, const double* gradient, int64_t K,
                          double* step) {
    double numSum = 0.0, denSum = 0.0;
    for (int64_t k = 0; k < K; ++k) {
        if (hDiag[k] == 0.0) return ORACLE_ZERO_DIVISION;
        numSum += gradient[k] / hDiag[k];
    }
    for (int64_t k = 0; k < K; ++k) denSum += 1.0 / hDiag[k];
    if (hConst == 0.0) return ORACLE_ZERO_DIVISION;
    double den = (1.0 / hConst) + denSum;
    if (den == 0.0) return ORACLE_ZERO_DIVISION;
    double b = numSum / den;
    for (int64_t k = 0; k < K; ++k) step[k] = (b - gradient[k]) / hDiag[k];
    return ORACLE_OK;
}

/* a9: getPredictedStepAlt — DE/dme.py:76-80 */
int oracle_predicted_step_alt(double hConst, const double* hDiag, const double* gradient,
                              const double* alphas, int64_t K, double* step) {
    double* x = (double*)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
    for (int64_t k = 0; k < K; ++k) x[k] = gradient[k] + alphas[k] * hDiag[k]; /* :77 */
    int rc = ORACLE_OK;
    double zs = 0.0, S = 0.0;
    for (int64_t k = 0; k < K && !rc; ++k) {
        if (x[k] == 0.0) rc = ORACLE_ZERO_DIVISION;
        else zs += alphas[k] / x[k];
    }
    if (!rc && hConst == 0.0) rc = ORACLE_ZERO_DIVISION;
    if (!rc) {
        double Z = 1.0 / hConst + zs;                                          /* :78 */
        for (int64_t k = 0; k < K; ++k) S += alphas[k] * gradient[k] / x[k];   /* :79 */
        if (Z == 0.0) rc = ORACLE_ZERO_DIVISION;
        else
            for (int64_t k = 0; k < K; ++k) step[k] = (S / Z - gradient[k]) / x[k]; /* :80 */
    }
    free(x);
    return rc;
}

/* a12: sqVectorSize — DE/dme.py:104 */
double oracle_sq_vector_size(const double* v, int64_t K) {
    double s = 0.0;
    for (int64_t k = 0; k < K; ++k) s += pow(v[k], 2.0);
    return s;
}

typedef struct {
    const double* Uflat;
    const int64_t* off;
    const double* V;
    int64_t K, lenV;
    double hyperprior;
    int64_t n_evals;
    double* eval_loss; /* optional, capacity eval_cap */
    int64_t eval_cap;
} ctx_t;

/* a11: testTrialPriors — DE/dme.py:97-102 */
static int test_trial(ctx_t* c, const double* trial, double* loss) {
    for (int64_t k = 0; k < c->K; ++k)
        if (trial[k] <= 0) { *loss = INFINITY; return ORACLE_OK; }
    int rc = oracle_total_loss(trial, c->K, c->Uflat, c->off, c->V, c->lenV, c->hyperprior, loss);
    if (!rc) {
        if (c->eval_loss && c->n_evals < c->eval_cap) c->eval_loss[c->n_evals] = *loss;
        c->n_evals++;
    }
    return rc;
}

/* a13: findDirichletPriors — DE/dme.py:126-190.
 * alpha[K] in: initAlphas, out: returned priors.  exit_reason: 1 = "Converged with small gradient"
 * (:145-147), 2 = "Converged with small learn rate" (:182-184), 3 = "Reached max iterations"
 * (:189-190).  trace_* (optional) record what the reference logs at :143 each iteration.
 * Returns ORACLE_OK or the code of the Python exception the reference would raise. */
int oracle_find_priors(const double* Uflat, const int64_t* off, const double* V, int64_t K,
                       int64_t lenV, double* alpha, int64_t iterations, double hyperprior,
                       double* trace_loss, double* trace_gradsq, double* trace_alpha,
                       int64_t trace_cap, int64_t* n_iters, int64_t* n_evals, int* exit_reason,
                       double* eval_loss, int64_t eval_cap) {
    ctx_t c = {Uflat, off, V, K, lenV, hyperprior, 0, eval_loss, eval_cap};
    size_t kb = sizeof(double) * (size_t)(K > 0 ? K : 1);
    double* priors = (double*)malloc(kb);
    double* gradient = (double*)malloc(kb);
    double* hdiag = (double*)malloc(kb);
    double* step = (double*)malloc(kb);
    double* trial = (double*)malloc(kb);
    memcpy(priors, alpha, sizeof(double) * (size_t)K);
    int rc = ORACLE_OK;
    int reason = 0;
    int64_t count = 0;
    double currentLoss, loss, hconst;

    /* :131 — getTotalLoss directly (no positivity test here) */
    rc = oracle_total_loss(priors, K, Uflat, off, V, lenV, hyperprior, &currentLoss);
    if (rc) goto done;
    if (c.eval_loss && c.n_evals < c.eval_cap) c.eval_loss[c.n_evals] = currentLoss;
    c.n_evals++;

    const double gradientToleranceSq = 0.0009765625;          /* 2 ** -10, :133 */
    const double learnRateTolerance = 9.5367431640625e-07;     /* 2 ** -20, :134 */

    while (count < iterations) { /* :137 */
        count += 1;
        rc = oracle_gradient(priors, K, Uflat, off, V, lenV, hyperprior, gradient); /* :141 */
        if (rc) goto done;
        double gradientSize = oracle_sq_vector_size(gradient, K); /* :142 */
        if (count - 1 < trace_cap) {
            if (trace_loss) trace_loss[count - 1] = currentLoss;
            if (trace_gradsq) trace_gradsq[count - 1] = gradientSize;
            if (trace_alpha) memcpy(trace_alpha + (count - 1) * K, priors, sizeof(double) * (size_t)K);
        }
        if (gradientSize < gradientToleranceSq) { reason = 1; goto done; } /* :145-147 */

        /* :150 predictStepUsingHessian (:86-89) */
        rc = oracle_hconst(priors, K, V, lenV, &hconst);
        if (rc) goto done;
        rc = oracle_hdiag(priors, K, Uflat, off, hdiag);
        if (rc) goto done;
        rc = oracle_predicted_step(hconst, hdiag, gradient, K, step);
        if (rc) goto done;
        for (int64_t i = 0; i < K; ++i) trial[i] = priors[i] + step[i]; /* :153-154 */

        rc = test_trial(&c, trial, &loss); /* :159 */
        if (rc) goto done;
        if (loss < currentLoss) { /* :160-163 */
            currentLoss = loss;
            memcpy(priors, trial, sizeof(double) * (size_t)K);
            continue;
        }

        /* :165 predictStepLogSpace (:91-94) recomputes both Hessian parts */
        rc = oracle_hconst(priors, K, V, lenV, &hconst);
        if (rc) goto done;
        rc = oracle_hdiag(priors, K, Uflat, off, hdiag);
        if (rc) goto done;
        rc = oracle_predicted_step_alt(hconst, hdiag, gradient, priors, K, step);
        if (rc) goto done;
        for (int64_t i = 0; i < K; ++i) { /* :166-171 */
            double e = exp(step[i]);
            /* math.exp raises OverflowError when the result overflows; bare except keeps prior */
            if (isinf(e) && !isinf(step[i])) trial[i] = priors[i];
            else trial[i] = priors[i] * e;
        }
        rc = test_trial(&c, trial, &loss); /* :172 */
        if (rc) goto done;

        double learnRate = 1.0; /* :175 */
        while (loss > currentLoss) { /* :176-180 */
            learnRate *= 0.9;
            for (int64_t i = 0; i < K; ++i) trial[i] = priors[i] + gradient[i] * learnRate;
            rc = test_trial(&c, trial, &loss);
            if (rc) goto done;
        }
        if (learnRate < learnRateTolerance) { reason = 2; goto done; } /* :182-184 */
        currentLoss = loss; /* :186-187 */
        memcpy(priors, trial, sizeof(double) * (size_t)K);
    }
    reason = 3; /* :189-190 */

done:
    if (!rc) memcpy(alpha, priors, sizeof(double) * (size_t)K);
    if (n_iters) *n_iters = count;
    if (n_evals) *n_evals = c.n_evals;
    if (exit_reason) *exit_reason = reason;
    free(priors); free(gradient); free(hdiag); free(step); free(trial);
    return rc;
}
