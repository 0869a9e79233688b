"""CPU tier: vector observations (SmallOutputLGC, p > 1, diagonal noise) run as p scalar micro-steps by the
engine (same headers as the HIP kernels, via tests/hostsim) against the oracle's JOINT update
(oracle/lgssm_ref.py: posterior_and_lml_small, lgc.jl:129-141)."""
import numpy as np
import pytest

from oracle import lgssm_ref as ref
from tests import _util as U


@pytest.mark.parametrize("d,p", [(2, 2), (3, 2), (4, 3), (3, 5)])
@pytest.mark.parametrize("tv", [True, False])
@pytest.mark.parametrize("ordering", ["F", "R"])
def test_vector_obs(d, p, tv, ordering):
    rng = np.random.default_rng(100 * d + 10 * p + tv)
    T = 41
    model = U.random_lgssm_small(rng, tv, d, p, T, ordering)
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    for L0, BS in [(2 * p, 3), (5 * p, 2)]:
        lp = ref.logpdf(model, y)
        r = U.hostsim_run(model, 0, y=y, L0=L0, BS=BS)
        assert r["rc"] == 0 and abs(r["lml"] - lp) <= 1e-10 * abs(lp)
        fm, fP = ref.filter_(model, y)
        r = U.hostsim_run(model, 1, y=y, L0=L0, BS=BS)
        np.testing.assert_allclose(r["m"], fm, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r["P"], fP, rtol=1e-9, atol=1e-10)
        mm, mC = ref.marginals(model)
        r = U.hostsim_run(model, 3, L0=L0, BS=BS)
        np.testing.assert_allclose(r["mean"], mm, rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(r["var"], np.diagonal(mC, axis1=-2, axis2=-1), rtol=1e-10, atol=1e-11)
        r = U.hostsim_run(model, 4, L0=L0, BS=BS, eps=eps)
        np.testing.assert_allclose(r["mean"], y, rtol=1e-9, atol=1e-9)
        if ordering == "F":
            post = ref.posterior(model, y)
            Rn = rng.random((T, p)) * 0.1
            post_n = ref.replace_observation_noise_cov(post, np.stack([np.diag(v) for v in Rn]))
            pm, pC = ref.marginals(post_n)
            r = U.hostsim_run(model, 2, y=y, L0=L0, BS=BS, Rnew=Rn, want_ggl=True)
            assert r["rc"] == 0
            np.testing.assert_allclose(r["G"], post["A"], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(r["L"], post["Q"], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(r["mean"], pm, rtol=1e-8, atol=1e-8)
            np.testing.assert_allclose(r["var"], np.diagonal(pC, axis1=-2, axis2=-1), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("per_element", [False, True])
def test_vector_obs_missing(per_element):
    rng = np.random.default_rng(9)
    T, d, p = 37, 3, 3
    model = U.random_lgssm_small(rng, True, d, p, T)
    y = rng.standard_normal((T, p))
    if per_element:
        missing = rng.random((T, p)) < 0.3                      # lgc.jl:143-151 (diagonal noise)
        mask = missing
    else:
        missing = rng.random(T) < 0.3                            # whole time steps (missings.jl:8-13)
        mask = np.repeat(missing[:, None], p, axis=1)
    lp = ref.logpdf_missing(model, y, missing)
    r = U.hostsim_run(model, 0, y=y, missing=mask, L0=2 * p, BS=3)
    assert abs(r["lml"] - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_missing(model, y, missing)
    r = U.hostsim_run(model, 1, y=y, missing=mask, L0=2 * p, BS=3)
    np.testing.assert_allclose(r["m"], fm, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["P"], fP, rtol=1e-9, atol=1e-10)
