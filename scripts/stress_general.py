#!/usr/bin/env python3
"""Randomised check of the general (chunked-scan) engine against the sequential C oracle: random stable models of d = 1..8 with EVERY
combination of shared / per-step blocks (A, a, Q, H, h, R each on its own), lengths around the chunk / workgroup / scan-level boundaries,
random chunk sizes; logpdf, filtering distributions, posterior marginals, prior marginals, rand.
usage: stress_general.py [n_cases] [seed]      (START=<case> END=<case> select cases; every case is reproducible on its own)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import seq_kalman as sk  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LENGTHS = [1, 2, 7, 63, 64, 65, 255, 257, 1000, 4097, 16385, 70_001]
bad = 0
for case in range(int(os.environ.get("START", "0")), min(n_cases, int(os.environ.get("END", "1000000")))):
    rng = np.random.default_rng([seed, case])
    d = int(rng.integers(1, 9))
    T = int(LENGTHS[rng.integers(len(LENGTHS))])
    per = {k: bool(rng.random() < 0.5) and T > 1 for k in "AaQHhR"}

    def psd(n, lo, hi):
        U = np.linalg.qr(rng.standard_normal((n, n)))[0]
        return (U * (rng.random(n) * (hi - lo) + lo)) @ U.T
    nA, na, nQ, nH, nh, nR = (T if per[k] else 1 for k in "AaQHhR")
    style = int(rng.integers(3))       # 0 generic, 1 slowly mixing (A close to I, small Q), 2 fast forgetting
    if style == 1:
        A = np.stack([np.eye(d) * (1 - 0.02 * rng.random()) + 0.02 * rng.standard_normal((d, d)) / d for _ in range(nA)])
        # (kept stable: an exploding state -- spectral radius above 1 over thousands of steps -- is outside what a GP's state-space form
        #  produces, and there the scan elements' products overflow the precision the sequential recursion keeps)
        A = np.stack([Ai / max(1.0, np.abs(np.linalg.eigvals(Ai)).max() / 0.9995) for Ai in A])
        Q = np.stack([psd(d, 1e-4, 1e-2) for _ in range(nQ)])
    elif style == 2:
        A = np.stack([0.1 * rng.standard_normal((d, d)) for _ in range(nA)])
        Q = np.stack([psd(d, 0.5, 2.0) for _ in range(nQ)])
    else:
        A = np.stack([-psd(d, 0.1, 0.9) + 0.2 * rng.standard_normal((d, d)) for _ in range(nA)])
        A = np.stack([Ai / max(1.0, 1.1 * np.abs(np.linalg.eigvals(Ai)).max()) for Ai in A])
        Q = np.stack([psd(d, 0.2, 1.5) for _ in range(nQ)])
    model = dict(ordering="F", kind="scalar", T=T, A=A, a=0.3 * rng.standard_normal((na, d)), Q=Q, H=rng.standard_normal((nH, d)),
                 h=rng.standard_normal(nh), R=np.exp(rng.uniform(np.log(1e-3), np.log(2.0), nR)), x0m=rng.standard_normal(d), x0P=psd(d, 0.5, 1.5))
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = sk.rand(model, *eps)
    Rn = np.exp(rng.uniform(np.log(1e-3), np.log(0.5), T if rng.random() < 0.5 else 1))
    chunk = int(rng.choice([0, 0, 1, 3, 17, 64]))
    lp_o, fm, fP = sk.filter_(model, y, want_states=True)
    pm, pv = sk.posterior_marginals(model, y, Rn)
    qm, qv = sk.prior_marginals(model)
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    if os.environ.get("CHUNK") is not None:
        chunk = int(os.environ["CHUNK"])
    if chunk:
        dm.handle().set_option(tgp._lib.OPT_CHUNK, chunk)
    if os.environ.get("VARIANT") is not None:
        dm.handle().set_option(tgp._lib.OPT_VARIANT, int(os.environ["VARIANT"]))
    if os.environ.get("GROUP") is not None:
        dm.handle().set_option(tgp._lib.OPT_GROUP, int(os.environ["GROUP"]))
    msgs = []

    def close(name, got, want, rtol=1e-8):
        got, want = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(want, dtype=np.float64).reshape(-1)
        sc = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
        if got.shape != want.shape or not np.all(np.abs(got - want) <= rtol * sc):
            msgs.append(f"{name}: max err {np.max(np.abs(got - want)) / sc if got.shape == want.shape else 'shape'}")
    try:
        lp = tgp.logpdf(dm, y)
        if not abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o)):
            msgs.append(f"logpdf {lp} vs {lp_o}")
        m, Pf = tgp._filter(dm, y)
        close("filter mean", m, fm)
        close("filter cov", Pf, fP)
        gm, gv = tgp.posterior_marginals(dm, y, Rn)
        close("posterior mean", gm, pm)
        close("posterior var", gv, pv)
        lp2, gm2, gv2 = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
        close("combined call", np.concatenate([[lp2], gm2, gv2]), np.concatenate([[lp_o], pm, pv]))
        um, uv = tgp.marginals(dm)
        close("prior mean", um, qm)
        close("prior var", uv, qv)
        close("rand", tgp.rand(eps, dm), y, 1e-7)
    except Exception as ex:      # noqa: BLE001
        msgs.append(f"{type(ex).__name__}: {ex}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={d} T={T} per-step={''.join(k for k in 'AaQHhR' if per[k]) or '-'} style={style} chunk={chunk} Rn={'T' if Rn.shape[0] > 1 else '1'} "
          f"variant={dm.handle().lib.tgp_kernel_variant(dm.handle().h)} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}")
