#!/usr/bin/env python3
"""Randomised check at LONG series (millions of steps: thousands of workgroups, the sequential head, the last tiles' tail tables): random sums of
scaled, stretched Matern terms (d <= 8), random spacing and noise; logpdf, posterior marginals and rand of the default engine against the
sequential C oracle.  usage: stress_long.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import components as oc  # noqa: E402
from oracle import seq_kalman as sk  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAMES = ["matern12", "matern32", "matern52"]
DIM = dict(matern12=1, matern32=2, matern52=3)
bad = 0
for case in range(n_cases):
    while True:
        terms = [(NAMES[rng.integers(3)], float(np.exp(rng.normal(0, 0.5))), float(np.exp(rng.normal(0, 0.6)))) for _ in range(rng.integers(1, 4))]
        if sum(DIM[t[0]] for t in terms) <= 8:
            break
    dt = float(np.exp(rng.uniform(np.log(0.02), np.log(0.5))))
    noise = float(np.exp(rng.uniform(np.log(1e-3), np.log(1.0))))
    T = int(rng.choice([1_000_003, 2_500_001, 4_194_304, 6_000_017]))
    spec = tuple(("scaled", s2, ("stretched", s, (nm,))) for nm, s2, s in terms)
    spec = spec[0] if len(spec) == 1 else ("sum",) + spec
    model = oc.build_lgssm(spec, ("regular", 0.0, dt, T), noise)
    d = len(model["x0m"])
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = sk.rand(model, *eps)
    Rn = np.array([float(np.exp(rng.normal(-2, 1)))])
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    msgs = []
    yr = tgp.rand(eps, dm)
    sc = max(1.0, float(np.max(np.abs(y))))
    if not np.max(np.abs(yr - y)) <= 1e-9 * sc:
        msgs.append(f"rand {np.max(np.abs(yr - y)):.2e}")
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    names = sorted(hd.profile())
    lp_ref = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, Rn)
    if not abs(lp - lp_ref) <= 1e-10 * abs(lp_ref):
        msgs.append(f"logpdf {lp} vs {lp_ref}")
    scale = max(1.0, float(np.max(np.abs(pm))))
    if not (np.max(np.abs(mean - pm)) <= 1e-8 * scale and np.max(np.abs(var - pv)) <= 1e-8 * max(1.0, float(np.max(pv)))):
        msgs.append(f"marginals {np.max(np.abs(mean - pm)):.2e} {np.max(np.abs(var - pv)):.2e}")
    tag = "FAIL" if msgs else "ok"
    bad += bool(msgs)
    print(f"[{case:3d}] {tag} d={d} T={T} dt={dt:.4f} noise={noise:.2e} terms={[(t[0][6:], round(t[1], 2), round(t[2], 2)) for t in terms]} kernels={[n for n in names if n.startswith(('k_steady', 'k_rand', 'k_apply', 'k_reduce_f'))][:4]} {'; '.join(msgs)}", flush=True)
    del dm
print(f"{bad} failing cases of {n_cases}")
