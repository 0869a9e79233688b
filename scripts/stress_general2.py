#!/usr/bin/env python3
"""Second randomised sweep of the general engine, against the oracle's literal restatement (oracle/lgssm_ref.py, Python loops: short series):
d = 1..16, scalar and vector observations (p = 1..5, diagonal noise), Forward and Reverse priors, every shared / per-step combination,
missing data (whole steps or single elements), random chunk sizes. logpdf, filtering distributions, prior marginals, rand, and for
Forward priors the posterior marginals.   usage: stress_general2.py [n_cases] [seed]   (START / END select cases)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import lgssm_ref as ref  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LENGTHS = [1, 2, 5, 63, 64, 65, 129, 300, 517]
bad = 0
for case in range(int(os.environ.get("START", "0")), min(n_cases, int(os.environ.get("END", "1000000")))):
    rng = np.random.default_rng([seed, case])
    d = int(rng.integers(int(os.environ.get("DMIN", "1")), int(os.environ.get("DMAX", "16")) + 1))        # (DMIN=17 DMAX=70: the dense engine)
    p = int(rng.integers(1, int(os.environ.get("PMAX", "5")) + 1)) if rng.random() < 0.6 else 1
    T = int(LENGTHS[rng.integers(len(LENGTHS))])
    ordering = "R" if rng.random() < 0.3 else "F"
    per = {k: bool(rng.random() < 0.5) and T > 1 for k in "AaQHhR"}

    def psd(n, lo, hi):
        U = np.linalg.qr(rng.standard_normal((n, n)))[0]
        return (U * (rng.random(n) * (hi - lo) + lo)) @ U.T
    nA, na, nQ, nH, nh, nR = (T if per[k] else 1 for k in "AaQHhR")
    A = np.stack([-psd(d, 0.1, 0.9) + 0.2 * rng.standard_normal((d, d)) for _ in range(nA)])
    A = np.stack([Ai / max(1.0, 1.1 * np.abs(np.linalg.eigvals(Ai)).max()) for Ai in A])
    Q = np.stack([psd(d, 0.2, 1.5) for _ in range(nQ)])
    small = p > 1 or rng.random() < 0.2
    if small:
        H, h = rng.standard_normal((nH, p, d)), rng.standard_normal((nh, p))
        Rd = rng.random((nR, p)) + 0.1
        R = np.stack([np.diag(v) for v in Rd])
    else:
        H, h, R = rng.standard_normal((nH, d)), rng.standard_normal(nh), rng.random(nR) + 0.1
    model = dict(ordering=ordering, kind="small" if small else "scalar", T=T, A=A, a=0.3 * rng.standard_normal((na, d)), Q=Q, H=H, h=h, R=R,
                 x0m=rng.standard_normal(d), x0P=psd(d, 0.9, 1.1))
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)) if small else rng.standard_normal(T), rng.standard_normal(d))
    y = np.asarray(ref.rand(model, *eps))
    miss_kind = int(rng.integers(3))          # 0 none, 1 whole time steps, 2 single elements (vector observations)
    missing = None
    if miss_kind == 1:
        missing = rng.random(T) < 0.25
    elif miss_kind == 2 and small and p > 1:
        missing = rng.random((T, p)) < 0.25
    ym = y.copy()
    if missing is not None:
        ym[missing] = np.nan
    chunk = int(rng.choice([0, 0, 1, 2, 7, 40]))
    tr = tgp.GaussMarkovModel(tgp.Forward if ordering == "F" else tgp.Reverse, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    em = tgp.SmallOutputLGC(H, h, Rd) if small else tgp.ScalarOutputLGC(H, h, R)
    dm = tgp.LGSSM(tr, em, T=T)
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
            msgs.append(f"{name}: max err {np.max(np.abs(got - want)) / sc if got.shape == want.shape else (got.shape, want.shape)}")
    try:
        if missing is None:
            lp_o, (fm, fP) = ref.logpdf(model, y), ref.filter_(model, y)
        else:
            lp_o, (fm, fP) = ref.logpdf_missing(model, y, missing), ref.filter_missing(model, y, missing)
        lp = tgp.logpdf(dm, ym)
        if not abs(lp - lp_o) <= 1e-10 * max(1.0, abs(lp_o)):
            msgs.append(f"logpdf {lp} vs {lp_o}")
        m, Pf = tgp._filter(dm, ym)
        close("filter mean", m, fm)
        close("filter cov", Pf, fP)
        mm, mC = ref.marginals(model)
        um, uv = tgp.marginals(dm)
        close("prior mean", um, mm)
        close("prior var", uv, np.diagonal(mC, axis1=-2, axis2=-1) if small else mC)
        close("rand", tgp.rand(eps, dm), y, 1e-7)
        if ordering == "F" and T > 1:
            post = ref.posterior(model, y) if missing is None else ref.posterior_missing(model, y, missing)
            Rn = rng.random((T, p)) * 0.1 + 1e-3 if small else rng.random(T) * 0.1 + 1e-3
            pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, np.stack([np.diag(v) for v in Rn]) if small else Rn))
            gm, gv = tgp.posterior_marginals(dm, ym, Rn)
            close("posterior mean", gm, pm)
            close("posterior var", gv, np.diagonal(pC, axis1=-2, axis2=-1) if small else pC)
    except Exception as ex:      # noqa: BLE001
        import traceback
        msgs.append(f"{type(ex).__name__}: {ex} @ {traceback.extract_tb(ex.__traceback__)[-1].lineno}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={d} p={p}{' small' if small else ''} {ordering} T={T} per-step={''.join(k for k in 'AaQHhR' if per[k]) or '-'} missing={miss_kind if missing is not None else 0} "
          f"chunk={chunk} variant={dm.handle().lib.tgp_kernel_variant(dm.handle().h)} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}")
