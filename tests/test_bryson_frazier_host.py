"""CPU tier: the adjoint (modified Bryson-Frazier) form the persistent mid-d backward pass runs (tgp_dense_fused.hpp), restated in
NumPy (oracle.lgssm_ref.bryson_frazier_marginals), against the oracle's literal reverse-time model (lgssm.jl:193-238, 99-115):
identical up to the effect of the reference's 1e-10 jitter on its own result."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref
from tests import _util as U


@pytest.mark.parametrize("d,tv", [(2, False), (5, True), (18, False), (24, True)])
def test_adjoint_form_equals_reverse_time_model_on_random_models(d, tv):
    rng = np.random.default_rng(50 + d)
    T = 120
    model = U.random_lgssm(rng, tv, d, T)
    y = rng.standard_normal(T)
    Rn = rng.random(T) * 0.1
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior(model, y), Rn))
    bm, bv = ref.bryson_frazier_marginals(model, y, Rn)
    np.testing.assert_allclose(bm, pm, rtol=0, atol=1e-9 * max(1.0, np.abs(pm).max()))
    np.testing.assert_allclose(bv, pv, rtol=1e-9, atol=1e-11)


def test_adjoint_form_with_missing_steps():
    rng = np.random.default_rng(7)
    T, d = 90, 6
    model = U.random_lgssm(rng, True, d, T)
    y = rng.standard_normal(T)
    missing = rng.random(T) < 0.3
    Rn = np.full(T, 0.05)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior_missing(model, y, missing), Rn))
    bm, bv = ref.bryson_frazier_marginals(model, y, Rn, missing=missing)
    np.testing.assert_allclose(bm, pm, rtol=0, atol=1e-9 * max(1.0, np.abs(pm).max()))
    np.testing.assert_allclose(bv, pv, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("case", [(("matern52",), ("regular", 0.0, 0.1, 1500), 0.1, 1e-8),
                                  (("scaled", 1.0, ("stretched", 1 / 2.3, ("matern52",))), ("regular", -5.0, 1e-2, 1500), 0.5, 5e-7)],
                         ids=["matern52-dt0.1", "bench-parametrisation-dt0.01"])
def test_jitter_effect_on_gp_models_is_what_separates_the_two(case):
    """On the BASELINE kernels the two forms differ by 1e-9 ... 6e-8 of the mean's scale (2e-7 of the variance's): the size of the reference's own jitter
    effect, largest for the bench parametrisation (dt = 0.01, l = 2.3) -- the reason the scan engine (d <= 16, 1e-8 bar against the
    oracle) keeps the jittered RTS form and only the mid-d persistent pass uses this one."""
    k, t, s2, bound = case
    model, y, _ = U.gp_case(k, t, s2, seed=3)
    Rn = np.full(model["T"], 1e-18)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior(model, y), Rn))
    bm, bv = ref.bryson_frazier_marginals(model, y, Rn)
    assert np.abs(bm - pm).max() <= bound * np.abs(pm).max()
    assert np.abs(bv - pv).max() <= bound * pv.max()
