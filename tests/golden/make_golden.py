#!/usr/bin/env python3
"""Generates tests/golden/lgssm_golden.npz: small seeded inputs + expected outputs of the hot path.

The reference (Julia) cannot run in this image and its test-suite holds no golden vectors for this path
(SURVEY.md 8c), so these vectors are produced by the oracle's literal NumPy restatement
(oracle/lgssm_ref.py, itself pinned by the reference tests' state-space == dense-GP identities). They freeze
the oracle's behaviour (CPU tier: oracle vs golden) and give the GPU tier fixed known-answer cases. The `*_mp` entries are
the same recursions evaluated in 50-digit arithmetic (oracle/lgssm_mp.py): they measure the fp64 oracle's own rounding error.
PARITY UNPINNED stays: none of this is output of the reference itself.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import components as oc   # noqa: E402
from oracle import lgssm_mp          # noqa: E402
from oracle import lgssm_ref as ref   # noqa: E402
from tests import _util as U          # noqa: E402

CASES = {
    # name: (builder)
    "matern32_cfg1_small": lambda: U.gp_case(("matern32",), ("regular", 0.0, 0.1, 64), 0.1, seed=1),
    "matern52_regular": lambda: U.gp_case(("matern52",), ("regular", 0.0, 0.1, 50), 0.1, seed=2),
    "sum52_32_regular": lambda: U.gp_case(("sum", ("matern52",), ("matern32",)), ("regular", 0.0, 0.1, 40), 0.1, seed=3),
    "bench_param_matern52": lambda: U.gp_case(("scaled", 1.0, ("stretched", 1 / 2.3, ("matern52",))), ("regular", -5.0, 1e-2, 64), 0.5, seed=4),
    "irregular_hetero": lambda: U.gp_case(("scaled", 1.5, ("stretched", 0.7, ("matern52",))),
                                           np.cumsum(np.random.default_rng(5).random(48) * 0.1 + 0.05),
                                           np.random.default_rng(6).random(48) * 0.2 + 0.05, seed=5, mean=("const", 3.0)),
}


def random_case(seed, tv, d, T, ordering):
    rng = np.random.default_rng(seed)
    model = U.random_lgssm(rng, tv, d, T, ordering)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    return model, ref.rand(model, *eps), eps


for d in (1, 2, 4, 6):
    CASES[f"random_tv_d{d}"] = (lambda d=d: random_case(100 + d, True, d, 33, "F"))
CASES["random_ti_d3_reverse"] = lambda: random_case(200, False, 3, 29, "R")
CASES["random_tv_d3_reverse"] = lambda: random_case(201, True, 3, 29, "R")


def main():
    out = {}
    for name, build in CASES.items():
        model, y, eps = build()
        T = model["T"]
        for k in ("A", "a", "Q", "H", "h", "R", "x0m", "x0P"):
            out[f"{name}/{k}"] = np.asarray(model[k])
        out[f"{name}/ordering"] = np.array(0 if model["ordering"] == "F" else 1)
        out[f"{name}/y"] = y
        out[f"{name}/eps_t"], out[f"{name}/eps_e"], out[f"{name}/eps_0"] = eps
        out[f"{name}/logpdf"] = np.array(ref.logpdf(model, y))
        fm, fP = ref.filter_(model, y)
        out[f"{name}/filter_m"], out[f"{name}/filter_P"] = fm, fP
        mm, mv = ref.marginals(model)
        out[f"{name}/marg_mean"], out[f"{name}/marg_var"] = mm, mv
        missing = np.random.default_rng(7).random(T) < 0.25
        out[f"{name}/missing"] = missing
        out[f"{name}/logpdf_missing"] = np.array(ref.logpdf_missing(model, y, missing))
        if model["ordering"] == "F":
            post = ref.posterior(model, y)
            out[f"{name}/post_G"], out[f"{name}/post_g"], out[f"{name}/post_L"] = post["A"], post["a"], post["Q"]
            out[f"{name}/post_x0m"], out[f"{name}/post_x0P"] = post["x0m"], post["x0P"]
            Rn = np.random.default_rng(8).random(T) * 0.1
            out[f"{name}/Rnew"] = Rn
            pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
            out[f"{name}/post_mean"], out[f"{name}/post_var"] = pm, pv
            pmm, pvm = ref.marginals(ref.replace_observation_noise_cov(ref.posterior_missing(model, y, missing), Rn))
            out[f"{name}/post_mean_missing"], out[f"{name}/post_var_missing"] = pmm, pvm
            # the same recursions in 50-digit arithmetic (oracle/lgssm_mp.py): what the fp64 oracle's rounding error is measured against
            xp = lgssm_mp.run(model, y, Rn)
            out[f"{name}/logpdf_mp"] = np.array(xp["logpdf"])
            out[f"{name}/post_mean_mp"], out[f"{name}/post_var_mp"] = xp["post_mean"], xp["post_var"]
            out[f"{name}/logpdf_missing_mp"] = np.array(lgssm_mp.run(model, y, missing=missing)["logpdf"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lgssm_golden.npz"), **out)
    print(f"wrote {len(CASES)} cases, {len(out)} arrays")


if __name__ == "__main__":
    main()
