#!/usr/bin/env python3
"""Randomised check of the separable space-time path (to_gauss_markov.jl: kron(I, A_t), kron(K_r, Q_t), vector observations of the Nr
space points per time step; d = Nr * d_t from 1 into the dense engine's range) and of the eigen-decoupled shortcut, against the dense GP
with the separable kernel on the observed grid points: regular / irregular times, noise equal or different across space, missing points.
usage: stress_space_time.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import components as oc  # noqa: E402
from oracle import dense_gp as dg  # noqa: E402
from temporalgps_jl_amd import lti_sde, space_time  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(int(os.environ.get("START", "0")), min(n_cases, int(os.environ.get("END", "1000000")))):
    rng = np.random.default_rng([seed, case])
    Nr = int(rng.integers(1, 9))
    T = int(rng.choice([1, 3, 20, 80]))
    kt_name = ["matern12", "matern32", "matern52"][rng.integers(3)]
    kt = ("scaled", float(np.exp(rng.normal(0, 0.4))), ("stretched", float(np.exp(rng.normal(0, 0.5))), (kt_name,)))
    ks = ("se",) if rng.random() < 0.5 else ("stretched", float(np.exp(rng.normal(0, 0.4))), (["matern32", "matern52"][rng.integers(2)],))
    r = np.sort(rng.standard_normal(Nr)) * 1.5
    regular = rng.random() < 0.5
    dt = float(np.exp(rng.uniform(np.log(0.05), np.log(0.6))))
    t = ("regular", 0.2, dt, T) if regular else np.cumsum(rng.random(T) * 2 * dt + 0.05 * dt)
    equal_noise = rng.random() < 0.5
    s2 = float(np.exp(rng.uniform(np.log(0.02), np.log(0.5)))) if equal_noise else rng.random((T, Nr)) * 0.3 + 0.02
    msgs = []

    def close(name, got, want, rtol, atol=1e-8):
        got, want = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(want, dtype=np.float64).reshape(-1)
        if got.shape != want.shape or not np.all(np.abs(got - want) <= rtol * np.maximum(1.0, np.abs(want)) + atol):
            msgs.append(f"{name}: max err {np.max(np.abs(got - want)) if got.shape == want.shape else (got.shape, want.shape)}")
    d = None
    try:
        tk = lambda spec: space_time.SEKernel() if spec == ("se",) else lti_sde.to_kernel(spec)
        tt = lti_sde.RegularSpacing(t[1], t[2], t[3]) if regular else t
        k, grid = space_time.Separable(tk(ks), tk(kt)), space_time.RectilinearGrid(r, tt)
        K = dg.separable_kernelmatrix(ks, kt, r, oc.times(t))
        noise = np.full(T * Nr, s2) if np.ndim(s2) == 0 else np.asarray(s2).reshape(-1)
        L = np.linalg.cholesky(K + np.diag(noise))
        y = L @ rng.standard_normal(T * Nr)
        Y = y.reshape(T, Nr)
        Ym = Y.copy()
        if T * Nr > 2 and rng.random() < 0.5:
            Ym[rng.random((T, Nr)) < 0.2] = np.nan
        obs = ~np.isnan(Ym).reshape(-1)
        lp_d = dg.mvn_logpdf((K + np.diag(noise))[np.ix_(obs, obs)], y[obs]) if obs.any() else 0.0
        dm = space_time.build_lgssm(k, grid, s2)
        d = dm.dim
        lp = tgp.logpdf(dm, Ym)
        if not abs(lp - lp_d) <= 1e-6 * max(1.0, abs(lp_d)):
            msgs.append(f"literal model logpdf {lp} vs dense GP {lp_d}")
        if T > 1 and obs.all():
            gm, gv = tgp.posterior_marginals(dm, Y, np.full((1, Nr), 0.1))
            mu_d, var_d = dg.mvn_posterior_marginals(K, noise, y, 0.1)
            close("literal model posterior mean", gm, mu_d, 1e-5, 1e-6)
            close("literal model posterior var", gv, var_d, 1e-5, 1e-6)
            if equal_noise:
                dec = space_time.DecoupledSpaceTime(k, grid, s2)
                lp2 = dec.logpdf(Y)
                if not abs(lp2 - lp_d) <= 1e-6 * max(1.0, abs(lp_d)):
                    msgs.append(f"decoupled logpdf {lp2} vs dense GP {lp_d}")
                m2, v2 = dec.posterior_marginals(Y, 0.1)
                close("decoupled posterior mean", m2, mu_d, 1e-5, 1e-6)
                close("decoupled posterior var", v2, var_d, 1e-5, 1e-6)
    except Exception as ex:      # noqa: BLE001
        import traceback
        msgs.append(f"{type(ex).__name__}: {ex} @ {traceback.extract_tb(ex.__traceback__)[-1].lineno}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={d} Nr={Nr} T={T} {'regular' if regular else 'irregular'} time={kt_name} space={ks[0]} noise={'equal' if equal_noise else 'per-point'} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}")
