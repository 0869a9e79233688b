"""CPU tier: the time-parallel algorithm (the exact headers the HIP kernels instantiate, run sequentially
by tests/hostsim) against the oracle's literal restatement of the reference's sequential recursions.
Tolerances (fp64): logpdf rel 1e-10; filter/posterior states abs 1e-9 * scale; rand rel 1e-9."""
import numpy as np
import pytest

from oracle import lgssm_ref as ref
from tests import _util as U

GP_CASES = [
    (("matern12",), ("regular", 0.0, 0.1, 97), 0.1),
    (("matern32",), ("regular", 0.0, 0.1, 131), 0.1),
    (("matern52",), ("regular", 0.0, 0.1, 64), 0.1),
    (("sum", ("matern52",), ("matern32",)), ("regular", 0.0, 0.1, 50), 0.1),
    (("sum", ("matern52",), ("matern52",)), ("regular", 0.0, 0.05, 40), 0.2),
    (("scaled", 1.0, ("stretched", 1 / 2.3, ("matern52",))), ("regular", -5.0, 1e-2, 200), 0.5),
]


@pytest.mark.parametrize("i", range(len(GP_CASES)))
@pytest.mark.parametrize("L0,BS", [(4, 3), (7, 2), (1, 4), (64, 256)])
def test_gp_lti(i, L0, BS):
    k, t, s2 = GP_CASES[i]
    model, y, eps = U.gp_case(k, t, s2, seed=i)
    _check_all(model, y, eps, L0, BS)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 8, 10, 16])
@pytest.mark.parametrize("tv", [True, False])
@pytest.mark.parametrize("hetero", [False, True])
def test_random_lgssm(d, tv, hetero):
    rng = np.random.default_rng(10 * d + tv)
    T = 53
    model = U.random_lgssm(rng, tv, d, T)
    if hetero and not tv:
        model["R"] = rng.random(T) + 0.1
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    _check_all(model, y, eps, 5, 3)


def _check_all(model, y, eps, L0, BS):
    T = model["T"]
    lp = ref.logpdf(model, y)
    r = U.hostsim_run(model, 0, y=y, L0=L0, BS=BS)
    assert r["rc"] == 0
    assert abs(r["lml"] - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_(model, y)
    r = U.hostsim_run(model, 1, y=y, L0=L0, BS=BS)
    np.testing.assert_allclose(r["m"], fm, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["P"], fP, rtol=1e-9, atol=1e-10)
    post = ref.posterior(model, y)
    Rn = np.random.default_rng(5).random(T) * 0.1
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    r = U.hostsim_run(model, 2, y=y, L0=L0, BS=BS, Rnew=Rn, want_ggl=True)
    assert r["rc"] == 0
    np.testing.assert_allclose(r["G"], post["A"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(r["g"], post["a"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(r["L"], post["Q"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(r["xfm"], post["x0m"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["xfP"], post["x0P"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["mean"], pm, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(r["var"], pv, rtol=1e-8, atol=1e-9)
    mm, mv = ref.marginals(model)
    r = U.hostsim_run(model, 3, L0=L0, BS=BS)
    np.testing.assert_allclose(r["mean"], mm, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(r["var"], mv, rtol=1e-10, atol=1e-11)
    r = U.hostsim_run(model, 4, L0=L0, BS=BS, eps=eps)
    np.testing.assert_allclose(r["mean"], y, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tv", [True, False])
def test_missing(tv):
    rng = np.random.default_rng(3)
    T, d = 41, 3
    model = U.random_lgssm(rng, tv, d, T)
    y = rng.standard_normal(T)
    missing = rng.random(T) < 0.3
    lp = ref.logpdf_missing(model, y, missing)
    r = U.hostsim_run(model, 0, y=y, missing=missing)
    assert abs(r["lml"] - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_missing(model, y, missing)
    r = U.hostsim_run(model, 1, y=y, missing=missing)
    np.testing.assert_allclose(r["m"], fm, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["P"], fP, rtol=1e-9, atol=1e-10)
    post = ref.posterior_missing(model, y, missing)
    Rn = np.zeros(T)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    r = U.hostsim_run(model, 2, y=y, missing=missing, Rnew=Rn)
    np.testing.assert_allclose(r["mean"], pm, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(r["var"], pv, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("tv", [True, False])
def test_reverse_ordering(tv):
    rng = np.random.default_rng(11)
    T, d = 37, 3
    model = U.random_lgssm(rng, tv, d, T, ordering="R")
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    lp = ref.logpdf(model, y)
    r = U.hostsim_run(model, 0, y=y)
    assert abs(r["lml"] - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_(model, y)
    r = U.hostsim_run(model, 1, y=y)
    np.testing.assert_allclose(r["m"], fm, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(r["P"], fP, rtol=1e-9, atol=1e-10)
    mm, mv = ref.marginals(model)
    r = U.hostsim_run(model, 3)
    np.testing.assert_allclose(r["mean"], mm, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(r["var"], mv, rtol=1e-10, atol=1e-11)
    r = U.hostsim_run(model, 4, eps=eps)
    np.testing.assert_allclose(r["mean"], y, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tv", [False, True])
@pytest.mark.parametrize("d", [1, 2, 3, 5])
def test_reverse_ordered_posterior_matches_oracle(tv, d):
    """step_posterior(::Reverse) (lgssm.jl:223-228): update, then predict, then invert_dynamics(xp, xf, t) -- the reference passes
    the predicted state in the filtered slot -- and the last step's trailing predict gives the posterior's x0. The engine's
    MODE 3 pass (same chunk code as the HIP kernels, run on the host) against the oracle's literal restatement."""
    rng = np.random.default_rng(40 + d + 10 * tv)
    T = 41
    model = U.random_lgssm(rng, tv, d, T, "R")
    y = rng.standard_normal(T)
    post = ref.posterior(model, y)
    for L0, BS in ((5, 3), (8, 4), (64, 3)):
        r = U.hostsim_run(model, 2, y=y, L0=L0, BS=BS, want_ggl=True)
        assert r["rc"] == 0
        np.testing.assert_allclose(r["G"], post["A"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r["g"], post["a"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r["L"], post["Q"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r["xfm"], post["x0m"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r["xfP"], post["x0P"], rtol=1e-10, atol=1e-12)
