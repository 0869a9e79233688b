#!/usr/bin/env python3
"""Randomised check of the GP-level interface (to_sde / logpdf / marginals / posterior at the training and at new inputs / posterior logpdf)
against the oracle's restatement (oracle/components.py) and the dense GP (oracle/dense_gp.py): random kernel EXPRESSIONS over Matern-1/2,
-3/2, -5/2, constant and cosine terms (scaled, stretched, summed, multiplied: d from 1 into the dense engine's range), regular and
irregular inputs, constant / custom means, homoscedastic / heteroscedastic noise, missing observations.
usage: stress_gp_api.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd  # noqa: E402,F401
from oracle import components as oc  # noqa: E402
from oracle import dense_gp as dg  # noqa: E402
from temporalgps_jl_amd import lti_sde as P  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
DIM = dict(matern12=1, matern32=2, matern52=3, constant=1, cosine=2)


def draw(rng, depth=0):
    """-> (spec, state dimension)"""
    u = rng.random()
    if depth >= 2 or u < 0.35:
        nm = ["matern12", "matern32", "matern52", "constant", "cosine"][rng.choice(5, p=[0.25, 0.3, 0.3, 0.075, 0.075])]
        return (nm,) if nm != "constant" else ("constant", float(np.exp(rng.normal(0, 0.5)))), DIM[nm]
    if u < 0.5:
        k, d = draw(rng, depth + 1)
        return ("scaled", float(np.exp(rng.normal(0, 0.7))), k), d
    if u < 0.65:
        k, d = draw(rng, depth + 1)
        return ("stretched", float(np.exp(rng.normal(0, 0.7))), k), d
    if u < 0.85:
        parts = [draw(rng, depth + 1) for _ in range(int(rng.integers(2, 4)))]
        return ("sum",) + tuple(p[0] for p in parts), sum(p[1] for p in parts)
    # (the factors of a product need an SDE form of their own -- base kernels, scaled / stretched: the reference multiplies to_sde.(k.kernels))
    parts = [chain(rng) for _ in range(int(rng.integers(2, 4)))]
    return ("product",) + tuple(p[0] for p in parts), int(np.prod([p[1] for p in parts]))


def chain(rng):
    nm = ["matern12", "matern32", "matern52", "constant", "cosine"][rng.choice(5, p=[0.3, 0.3, 0.25, 0.075, 0.075])]
    k, d = ((nm,) if nm != "constant" else ("constant", float(np.exp(rng.normal(0, 0.5))))), DIM[nm]
    for _ in range(int(rng.integers(0, 3))):
        k = ("scaled", float(np.exp(rng.normal(0, 0.5))), k) if rng.random() < 0.5 else ("stretched", float(np.exp(rng.normal(0, 0.5))), k)
    return k, d


bad = 0
for case in range(int(os.environ.get("START", "0")), min(n_cases, int(os.environ.get("END", "1000000")))):
    rng = np.random.default_rng([seed, case])
    while True:
        spec, d = draw(rng)
        if d <= 40:
            break
    N = int(rng.choice([1, 2, 17, 60, 150]))
    regular = rng.random() < 0.4
    dt = float(np.exp(rng.uniform(np.log(0.02), np.log(1.0))))
    x = P.RegularSpacing(0.3, dt, N) if regular else np.cumsum(rng.random(N) * 2 * dt + 0.02 * dt) - 1.0
    xs = x.collect() if regular else x
    s2 = float(np.exp(rng.uniform(np.log(0.05), np.log(1.0)))) if rng.random() < 0.5 else rng.random(N) * 0.5 + 0.05
    mk = int(rng.integers(3))
    mean_o = [None, ("const", 1.7), ("custom", lambda t: 0.5 * t - 1.0)][mk]
    mean_p = [None, P.ConstMean(1.7), P.CustomMean(lambda t: 0.5 * t - 1.0)][mk]
    msgs = []

    def close(name, got, want, rtol, atol=1e-9):
        got, want = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(want, dtype=np.float64).reshape(-1)
        if got.shape != want.shape or not np.all(np.abs(got - want) <= rtol * np.maximum(1.0, np.abs(want)) + atol):
            msgs.append(f"{name}: max err {np.max(np.abs(got - want)) if got.shape == want.shape else (got.shape, want.shape)}")
    try:
        k = P.to_kernel(spec)
        gp = P.GP(k) if mean_p is None else P.GP(mean_p, k)
        f = P.to_sde(gp, P.HIPStorage())
        fx = f(x, s2)
        y = P.rand(rng, fx)
        ym = y.copy()
        if N > 3 and rng.random() < 0.5:
            ym[rng.random(N) < 0.2] = np.nan
        keep = ~np.isnan(ym)
        lp = P.logpdf(fx, ym)
        s2v = np.broadcast_to(np.asarray(s2, dtype=np.float64), (N,))
        lp_o = oc.gp_logpdf(spec, x if not regular else ("regular", 0.3, dt, N), s2, y, mean_o, ~keep)
        if not abs(lp - lp_o) <= 1e-9 * max(1.0, abs(lp_o)):
            msgs.append(f"logpdf {lp} vs oracle {lp_o}")
        if keep.any():
            lp_d = dg.logpdf(spec, xs[keep], s2v[keep], y[keep], mean_o)
            if not abs(lp - lp_d) <= 1e-6 * max(1.0, abs(lp_d)):
                msgs.append(f"logpdf {lp} vs dense GP {lp_d}")
        m, sd = P.marginals(fx)
        md, vd = dg.marginals(spec, xs, s2v, mean_o)
        close("prior mean", m, md, 1e-7)
        close("prior var", sd ** 2, vd, 1e-7)
        if N >= 2 and keep.sum() >= 1:
            fpost = P.posterior(f(xs[keep], s2v[keep]), y[keep])
            M = int(rng.integers(1, 12))
            x_pr = np.sort(rng.random(M)) * (xs[-1] - xs[0] + 2 * dt) + xs[0] - dt
            s_pr = rng.random(M) * 0.2 + 0.05
            pm, psd = P.marginals(fpost(x_pr, s_pr))
            pmd, pvd = dg.posterior_marginals(spec, xs[keep], s2v[keep], y[keep], x_pr, s_pr, mean_o)
            close("posterior mean at new inputs", pm, pmd, 1e-5, 1e-6)
            close("posterior var at new inputs", psd ** 2, pvd, 1e-5, 1e-6)
            pm2, psd2 = P.marginals(fpost(xs[keep], 0.1))
            pmd2, pvd2 = dg.posterior_marginals(spec, xs[keep], s2v[keep], y[keep], xs[keep], 0.1, mean_o)
            close("posterior mean at the training inputs", pm2, pmd2, 1e-5, 1e-6)
            close("posterior var at the training inputs", psd2 ** 2, pvd2, 1e-5, 1e-6)
            y_pr = rng.standard_normal(M)
            lpp = P.logpdf(fpost(x_pr, s_pr), y_pr)
            lpp_d = dg.posterior_logpdf(spec, xs[keep], s2v[keep], y[keep], x_pr, s_pr, y_pr, mean_o)
            if not abs(lpp - lpp_d) <= 1e-7 * max(1.0, abs(lpp_d)):
                msgs.append(f"posterior logpdf {lpp} vs dense GP {lpp_d}")
            lpp_m = fpost(x_pr, s_pr)._logpdf_merged(y_pr)        # the reference's chain spelled out: the evaluated posterior over the joined inputs
            if not abs(lpp - lpp_m) <= 1e-5 * max(1.0, abs(lpp_d)):
                msgs.append(f"posterior logpdf {lpp} vs the evaluated-posterior route {lpp_m}")
            y_same = rng.standard_normal(int(keep.sum()))
            if rng.random() < 0.3:
                y_same[rng.random(y_same.shape[0]) < 0.2] = np.nan
            s_same = 0.1 if rng.random() < 0.5 else rng.random(y_same.shape[0]) * 0.2 + 0.02
            ks = ~np.isnan(y_same)
            if ks.any():
                lps = P.logpdf(fpost(xs[keep], s_same), y_same)       # at the training inputs: the pair statistic
                lps_d = dg.posterior_logpdf(spec, xs[keep], s2v[keep], y[keep], xs[keep][ks], np.broadcast_to(s_same, ks.shape)[ks], y_same[ks], mean_o)
                if not abs(lps - lps_d) <= 1e-7 * max(1.0, abs(lps_d)):
                    msgs.append(f"posterior logpdf at the training inputs {lps} vs dense GP {lps_d}")
    except Exception as ex:      # noqa: BLE001
        import traceback
        msgs.append(f"{type(ex).__name__}: {ex} @ {traceback.extract_tb(ex.__traceback__)[-1].lineno}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={d} N={N} {'regular' if regular else 'irregular'} dt={dt:.3f} mean={mk} noise={'scalar' if np.ndim(s2) == 0 else 'per-step'} spec={spec} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}")
