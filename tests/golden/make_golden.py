#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container, where /root/reference is mounted.  It imports the
reference's three importable restatements of the hot path

    DE = DirichletEstimation/dirichletMultinomialEstimation.py   (canonical, pure math)
    NP = NumpyPort/dirichletMultinomialEstimation.py             (numpy restatement)
    CPT = ConjugatePriorTools/dirichletMultinomialEstimation.py  (original, (Beta,W) prior)

feeds them inputs (the reference's own sample files + seeded synthetic matrices), and writes
the inputs and the reference's outputs as data-only fixtures (JSON + npz).  Nothing of the
reference's source text is stored.  The GPU box never sees /root/reference; tests read only
the fixtures written here.

Usage:  python tests/golden/make_golden.py            (rewrites every fixture)
"""
import importlib.util
import json
import logging
import math
import os
import random
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


DE = load_module("ref_de_dme", f"{REF}/DirichletEstimation/dirichletMultinomialEstimation.py")
NP = load_module("ref_np_dme", f"{REF}/NumpyPort/dirichletMultinomialEstimation.py")
CPT = load_module("ref_cpt_dme", f"{REF}/ConjugatePriorTools/dirichletMultinomialEstimation.py")
DE_SAMPLE = load_module("ref_de_sampling", f"{REF}/DirichletEstimation/samplingTools.py")


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def read_rows(path, delimiter):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line == "":
                continue
            rows.append([int(c) for c in line.split(delimiter)])
    return rows


def cli_init(rows, K):
    """DE/findDirichletPrior.py:70,87-89,104-108 executed literally."""
    priors = [0.0] * K
    for data in rows:
        sumData = sum(data)
        weightForMean = 1.0 / (1.0 + sumData)
        for i in range(0, K):
            priors[i] += data[i] * weightForMean
    acc = list(priors)
    priorSum = sum(priors) + 0.01
    for i in range(0, K):
        priors[i] /= priorSum
        priors[i] += 0.01
    return acc, priors


def build_stats(mod, rows, K):
    obj = mod.CompressedRowData(K)
    for r in rows:
        obj.appendRow(r, 1)
    return obj


class TraceHandler(logging.Handler):
    """Captures the reference's own per-iteration debug line (DE/dme.py:143) and exit lines.
    The reference formats the line with % before logging it, so the numbers are parsed back
    out of the text (float repr round-trips exactly); the priors vector is taken from a spy
    on priorGradient instead."""

    PAT = re.compile(r"^Iteration: (\d+) Loss: (\S+) ,Priors: .*, Gradient Size: (\S+)$", re.S)

    def __init__(self):
        super().__init__(level=logging.DEBUG)
        self.iters = []
        self.exit = None

    def emit(self, record):
        msg = record.getMessage()
        m = self.PAT.match(msg)
        if m:
            self.iters.append({"count": int(m.group(1)), "loss": float(m.group(2)),
                               "gradSq": float(m.group(3))})
        elif msg == "Converged with small gradient":
            self.exit = "A"
        elif msg == "Converged with small learn rate":
            self.exit = "B"
        elif msg == "Reached max iterations":
            self.exit = "C"


def traced_solve(mod, data, init, iterations, hyperprior):
    """Runs mod.findDirichletPriors and records the iteration trace plus every loss
    evaluation (trial point + value) in call order — i.e. the full branch history."""
    evals = []
    grads_at = []
    orig = mod.getTotalLoss
    orig_grad = mod.priorGradient

    def spy(trialPriors, d, hp=0):
        v = orig(trialPriors, d, hp)
        evals.append({"alpha": [float(x) for x in trialPriors], "loss": float(v)})
        return v

    def spy_grad(priors, d, hp=0):
        grads_at.append([float(x) for x in priors])
        return orig_grad(priors, d, hp)

    h = TraceHandler()
    root = logging.getLogger()
    old_level = root.level
    root.addHandler(h)
    root.setLevel(logging.DEBUG)
    mod.getTotalLoss = spy
    mod.priorGradient = spy_grad
    try:
        if mod is NP:
            init = np.asarray(init, dtype=float)
        out = mod.findDirichletPriors(data, init, iterations, hyperprior)
    finally:
        mod.getTotalLoss = orig
        mod.priorGradient = orig_grad
        root.removeHandler(h)
        root.setLevel(old_level)
    assert len(grads_at) == len(h.iters)
    for it, a in zip(h.iters, grads_at):
        it["priors"] = a
    out = [float(x) for x in out]
    return out, h.iters, h.exit, evals


def func_values(mod, data, alpha, hyperprior):
    """Function-level golden values a4..a9 at one alpha."""
    logprob_name = "dirichLogProb" if hasattr(mod, "dirichLogProb") else "dirichletLogProb"
    a = np.asarray(alpha, dtype=float) if mod is NP else list(alpha)
    grad = mod.priorGradient(a, data, hyperprior)
    hconst = mod.priorHessianConst(a, data)
    hdiag = mod.priorHessianDiag(a, data)
    out = {
        "alpha": [float(x) for x in alpha],
        "hyperprior": hyperprior,
        "logprob": float(getattr(mod, logprob_name)(a, data, hyperprior)),
        "loss": float(mod.getTotalLoss(a, data, hyperprior)),
        "grad": [float(x) for x in grad],
        "hconst": float(hconst),
        "hdiag": [float(x) for x in hdiag],
        "gradSq": float(mod.sqVectorSize(grad)),
    }
    try:
        out["step"] = [float(x) for x in mod.getPredictedStep(hconst, hdiag, grad)]
        out["stepAlt"] = [float(x) for x in mod.getPredictedStepAlt(hconst, hdiag, grad, a)]
    except ZeroDivisionError:
        out["step"] = None
        out["stepAlt"] = None
    return out


def stats_dict(obj):
    return {
        "U": [[int(x) for x in u] for u in obj.U],
        "V": [int(x) for x in obj.V],
    }


def case_from_rows(name, rows, K, inits, hyperpriors=(0,), iterations=50, store_rows=True,
                   extra_eval_alphas=(), keep_evals=True, variants=("DE", "NP")):
    """One fixture: stats + CLI init + solves for each (init, hyperprior, variant)."""
    de_obj = build_stats(DE, rows, K)
    fx = {"name": name, "K": K, "N": len(rows)}
    if store_rows:
        fx["rows"] = rows
    fx.update(stats_dict(de_obj))
    np_obj = build_stats(NP, rows, K)
    assert [list(map(int, u)) for u in np_obj.U] == fx["U"], "NP/DE U mismatch"
    assert list(map(int, np_obj.V)) == fx["V"], "NP/DE V mismatch"
    acc, alpha0 = cli_init(rows, K)
    fx["init_acc"] = acc
    fx["alpha0_cli"] = alpha0
    fx["solves"] = []
    for init_name, init in inits:
        if init == "cli":
            init = alpha0
        for hp in hyperpriors:
            for variant in variants:
                mod, obj = (DE, de_obj) if variant == "DE" else (NP, np_obj)
                rec = {"variant": variant, "init_name": init_name, "init": list(init),
                       "hyperprior": hp, "iterations": iterations}
                try:
                    init_arg = list(init)
                    out, iters, ex, evals = traced_solve(mod, obj, init_arg, iterations, hp)
                    rec.update({"final": out, "exit": ex, "n_iters": len(iters),
                                "n_evals": len(evals), "trace": iters,
                                "final_loss": float(mod.getTotalLoss(
                                    np.asarray(out) if mod is NP else out, obj, hp))})
                    if keep_evals:
                        rec["evals"] = evals
                except ZeroDivisionError as e:
                    rec.update({"raises": "ZeroDivisionError"})
                fx["solves"].append(rec)
    # function-level values (DE + NP) at alpha0, at the DE final of the first solve, and extras
    pts = [alpha0]
    for s in fx["solves"]:
        if s["variant"] == "DE" and "final" in s:
            pts.append(s["final"])
            break
    pts.extend(extra_eval_alphas)
    fx["funcs"] = []
    for a in pts:
        for hp in hyperpriors:
            fx["funcs"].append({"DE": func_values(DE, de_obj, a, hp),
                                "NP": func_values(NP, np_obj, a, hp)})
    return fx


def dump(fx, fname):
    path = os.path.join(OUT, fname)
    with open(path, "w") as f:
        json.dump(fx, f, separators=(",", ":"))
    print(f"wrote {fname}: {os.path.getsize(path)/1024:.1f} KiB")


# --------------------------------------------------------------------------------------
# synthetic matrices (seeded numpy; the matrix itself is stored in the fixture)
# --------------------------------------------------------------------------------------
def synth_dirmult(seed_alpha, seed, N, K, nlo, nhi, shape=2.0, scale=0.05, base=0.01):
    """SURVEY 8(d) C3-style generator at small N (numpy twin; the matrix is what is pinned)."""
    ra = np.random.default_rng(seed_alpha)
    alpha = base + ra.gamma(shape, scale, size=K)
    r = np.random.default_rng(seed)
    X = np.zeros((N, K), dtype=np.int64)
    for i in range(N):
        g = r.standard_gamma(alpha)
        if g.sum() <= 0:
            g = np.full(K, 1.0)
        p = g / g.sum()
        n = int(r.integers(nlo, nhi + 1))
        X[i] = r.multinomial(n, p)
    return alpha, X


def synth_heavy(seed, N, K, cap=100000):
    """Pareto row totals, alpha = 0.05 + Gamma(2, 0.3)."""
    r = np.random.default_rng(seed)
    alpha = 0.05 + r.gamma(2.0, 0.3, size=K)
    X = np.zeros((N, K), dtype=np.int64)
    for i in range(N):
        n = int(min(math.floor(5 * (r.pareto(1.2) + 1)), cap))
        p = r.dirichlet(alpha)
        X[i] = r.multinomial(n, p)
    return alpha, X


def main():
    logging.basicConfig(level=logging.CRITICAL)

    # 1-3: the reference's own sample files --------------------------------------------
    rows = read_rows(f"{REF}/ConjugatePriorTools/test.csv", "\t")
    assert rows == read_rows(f"{REF}/NumpyPort/test.csv", ",")
    fx = case_from_rows(
        "test_csv", rows, 2,
        inits=[("cli", "cli"), ("ones", [1.0, 1.0]), ("fifty", [50.0, 50.0]),
               ("tiny", [1e-3, 1e-3])],
        hyperpriors=(0, 0.25, -0.1),
        extra_eval_alphas=[[1.0, 1.0], [0.3, 2.5], [50.0, 50.0]])
    dump(fx, "test_csv.json")

    rows = read_rows(f"{REF}/DirichletEstimation/test.txt", "\t")
    fx = case_from_rows("test_txt", rows, 2, inits=[("cli", "cli"), ("ones", [1.0, 1.0])])
    dump(fx, "test_txt.json")

    rows = read_rows(f"{REF}/ConjugatePriorTools/oneDoublesided.csv", "\t")
    fx = case_from_rows("one_doublesided", rows, 2,
                        inits=[("cli", "cli"), ("ones", [1.0, 1.0])])
    dump(fx, "one_doublesided.json")

    # 4: all-zero column -> ZeroDivisionError (DE/dme.py:69) ----------------------------
    rows = [[3, 0, 2], [1, 0, 4], [2, 0, 2]]
    fx = case_from_rows("zero_column", rows, 3, inits=[("cli", "cli")])
    assert fx["solves"][0].get("raises") == "ZeroDivisionError"
    dump(fx, "zero_column.json")

    # 5: flipCoins-style via the reference's own samplers, seeded ------------------------
    random.seed(20260001)
    rows = []
    for i in range(1500):
        p = DE_SAMPLE.drawFromDirichlet([0.7, 1.2])
        rows.append(DE_SAMPLE.sampleFromMultinomial(p, 60))
    fx = case_from_rows("flipcoins_1500", rows, 2, inits=[("cli", "cli"), ("ones", [1.0, 1.0])])
    fx["alpha_true"] = [0.7, 1.2]
    dump(fx, "flipcoins_1500.json")

    # 6: K=20 heavy tail (C5-style, small N) ---------------------------------------------
    alpha, X = synth_heavy(20260005, 3000, 20)
    rows = [[int(v) for v in r] for r in X]
    fx = case_from_rows("heavy_k20", rows, 20, store_rows=False,
                        inits=[("cli", "cli"), ("ones", [1.0] * 20)], keep_evals=False)
    fx["alpha_true"] = [float(a) for a in alpha]
    np.savez_compressed(os.path.join(OUT, "heavy_k20_X.npz"), X=X.astype(np.int32))
    dump(fx, "heavy_k20.json")

    # 7: K=1000 small N (C3-style) ---------------------------------------------------------
    alpha, X = synth_dirmult(0, 20260003, 1500, 1000, 1, 500)
    # the reference divides by zero on an all-zero column; the C3 generator must keep every
    # column populated, so add one synthetic row per empty column (SURVEY 8d, C4 note).
    empty = np.where(X.sum(axis=0) == 0)[0]
    if len(empty):
        extra = np.zeros((len(empty), 1000), dtype=np.int64)
        extra[np.arange(len(empty)), empty] = 1
        X = np.concatenate([X, extra], axis=0)
    rows = [[int(v) for v in r] for r in X]
    fx = case_from_rows("dirmult_k1000", rows, 1000, store_rows=False,
                        inits=[("cli", "cli")], keep_evals=False)
    fx["alpha_true"] = [float(a) for a in alpha]
    fx["n_padded_rows"] = int(len(empty))
    np.savez_compressed(os.path.join(OUT, "dirmult_k1000_X.npz"), X=X.astype(np.uint16))
    dump(fx, "dirmult_k1000.json")

    # 8: K=50 mid-size with ragged U and lambda != 0 --------------------------------------
    alpha, X = synth_dirmult(7, 20260007, 4000, 50, 1, 100, shape=2.0, scale=0.4, base=0.05)
    rows = [[int(v) for v in r] for r in X]
    fx = case_from_rows("dirmult_k50", rows, 50, store_rows=False,
                        inits=[("cli", "cli"), ("ones", [1.0] * 50)],
                        hyperpriors=(0, 0.01), keep_evals=False)
    fx["alpha_true"] = [float(a) for a in alpha]
    np.savez_compressed(os.path.join(OUT, "dirmult_k50_X.npz"), X=X.astype(np.int32))
    dump(fx, "dirmult_k50.json")

    # 10: weighted rows (appendRow(row, w), DE/dme.py:113-123 with fractional w, as the reference's
    #     Dirichlet-mixture EM does, CPT/dirichletMixtureEstimation.py:129-133) ------------------
    rows = read_rows(f"{REF}/ConjugatePriorTools/test.csv", "\t")[:300]
    rw = np.random.default_rng(20260010)
    weights = [float(w) for w in rw.random(len(rows))]
    weights[5] = 0.0     # a zero-weight row still lengthens U / V
    rows[5] = [70, 3]
    wobj = DE.CompressedRowData(2)
    for r, w in zip(rows[:100], weights[:100]):
        wobj.appendRow(r, w)
    for r in rows[100:200]:
        wobj.appendRow(r, 1)          # mixed with unit-weight rows
    for r, w in zip(rows[200:], weights[200:]):
        wobj.appendRow(r, w)
    init = [1.0, 1.0]
    out, iters, ex, evals = traced_solve(DE, wobj, list(init), 50, 0)
    dump({"name": "weighted_rows", "K": 2, "rows": rows, "weights": weights,
          "unit_weight_span": [100, 200],
          "U": [[float(x) for x in u] for u in wobj.U], "V": [float(x) for x in wobj.V],
          "init": init, "final": out, "exit": ex, "n_iters": len(iters), "n_evals": len(evals),
          "final_loss": float(DE.getTotalLoss(out, wobj, 0)),
          "trace": iters}, "weighted_rows.json")

    # 9: CPT variant agrees with DE on test.csv with no hyperprior (third oracle) ---------
    rows = read_rows(f"{REF}/ConjugatePriorTools/test.csv", "\t")
    cobj = build_stats(CPT, rows, 2)
    _, a0 = cli_init(rows, 2)
    out = CPT.findDirichletPriors(cobj, list(a0), 50)
    dump({"name": "cpt_test_csv", "final": [float(x) for x in out],
          "loss": float(CPT.getTotalLoss(out, cobj))}, "cpt_test_csv.json")


if __name__ == "__main__":
    main()
