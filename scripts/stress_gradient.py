#!/usr/bin/env python3
"""Randomised check of logpdf_and_gradient: random sums of scaled / stretched Matern terms with a constant mean, (i) regular spacing -- the
adjoint pass, the tangent scans and central differences of the device logpdf must agree; (ii) irregular spacing (d <= 4) -- the dual-number
passes over the tiled SDE record against central differences.   usage: stress_gradient.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from temporalgps_jl_amd import lti_sde as P  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
NAMES = ["matern12", "matern32", "matern52"]
DIM = dict(matern12=1, matern32=2, matern52=3)
bad = 0
for case in range(int(os.environ.get("START", "0")), min(n_cases, int(os.environ.get("END", "1000000")))):
    rng = np.random.default_rng([seed, case])
    irregular = rng.random() < 0.5
    dmax = 4 if irregular else 6
    while True:
        terms = [(NAMES[rng.integers(3)], float(np.exp(rng.normal(0, 0.5))), float(np.exp(rng.normal(0, 0.5)))) for _ in range(rng.integers(1, 4))]
        if sum(DIM[t[0]] for t in terms) <= dmax:
            break
    d = sum(DIM[t[0]] for t in terms)
    T = int(rng.choice([30, 700, 5000, 40_000]))
    dt = float(np.exp(rng.uniform(np.log(0.03), np.log(0.5))))
    x = np.cumsum(rng.random(T) * 2 * dt + 0.05 * dt) if irregular else P.RegularSpacing(0.0, dt, T)
    noise = float(np.exp(rng.uniform(np.log(0.02), np.log(1.0))))
    ks = [P.ScaledKernel(v, P.StretchedKernel(s, P.to_kernel((nm,)))) for nm, v, s in terms]
    k = ks[0]
    for kk in ks[1:]:
        k = k + kk
    gp = P.GP(P.ConstMean(float(rng.normal())), k) if rng.random() < 0.5 else P.GP(k)
    msgs = []
    try:
        fx = P.to_sde(gp, P.HIPStorage())(x, noise)
        y = P.rand(rng, fx)
        lp_f, g_fd = P.logpdf_and_gradient(fx, y, method="fd")
        sc = max(1.0, max(abs(v) for v in g_fd.values()))
        methods = ["tangent"] if irregular else ["adjoint", "tangent"]
        for mth in methods:
            try:
                lp, g = P.logpdf_and_gradient(fx, y, method=mth) if not irregular else P.logpdf_and_gradient(fx, y)
            except tgp._lib.Unsupported as ex:
                print(f"      ({mth}: {ex})")
                continue
            if not abs(lp - lp_f) <= 1e-10 * max(1.0, abs(lp_f)):
                msgs.append(f"{mth} logpdf {lp} vs {lp_f}")
            err = max(abs(g[n] - g_fd[n]) for n in g_fd) / sc
            if set(g) != set(g_fd) or not err <= 2e-5:
                msgs.append(f"{mth} gradient vs central differences: {err:.2e} ({ {n: (g[n], g_fd[n]) for n in g_fd} })")
    except Exception as ex:      # noqa: BLE001
        import traceback
        msgs.append(f"{type(ex).__name__}: {ex} @ {traceback.extract_tb(ex.__traceback__)[-1].lineno}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={d} T={T} {'irregular' if irregular else 'regular'} dt={dt:.3f} noise={noise:.3f} terms={[(t[0][6:], round(t[1], 2), round(t[2], 2)) for t in terms]} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}")
