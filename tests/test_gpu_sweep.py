"""GPU tier (-m gpu): the sweep engine (tgp_sweep.hip, TGP_OPT_SWEEP) -- models whose gains vary in time: a missing-data mask, a
noise variance / emission offset per step, irregular spacing -- through the C ABI against the oracle (lgssm.jl:147-238 with missings.jl:8-41,
lti_sde.jl:135-146).  Tolerances as everywhere: logpdf 1e-10 relative, posterior marginals 1e-8 of their scale.  Every test asserts that the
sweep engine served the call (tgp_sweep_info) -- or, where it must decline, that it did."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk
from tests import _util as U

pytestmark = pytest.mark.gpu

KERNELS = [
    (("matern12",), 0.1, 0.1),
    (("matern32",), 0.1, 0.1),
    (("matern52",), 0.1, 0.1),
    (("sum", ("matern32",), ("matern12",)), 0.1, 0.2),
    (("sum", ("matern32",), ("stretched", 0.7, ("matern32",))), 0.15, 0.1),
    (("scaled", 1.3, ("stretched", 1 / 2.3, ("matern52",))), 0.05, 0.5),
]


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def _lti_device_model(tgp, model):
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    return tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])


def _reference(model, y, missing, Rn, fast=False):
    T = model["T"]
    Rn = np.broadcast_to(Rn, (T,)).copy()
    if fast:      # the C restatement (missing steps as the reference has them: y := 0, R := 1e15, compensated volume)
        m2 = model
        y0 = y
        comp = 0.0
        if missing is not None:
            R = np.broadcast_to(np.atleast_1d(model["R"]), (T,)).copy()
            R[missing] = 1e15
            m2 = dict(model, R=R)
            y0 = np.where(missing, 0.0, y)
            comp = ref.volume_compensation(int(missing.sum()))
        lp = sk.logpdf(m2, y0) + comp
        pm, pv = sk.posterior_marginals(m2, y0, Rn)
        return lp, pm, pv
    if missing is not None:
        lp = ref.logpdf_missing(model, y, missing)
        post = ref.posterior_missing(model, y, missing)
    else:
        lp = ref.logpdf(model, y)
        post = ref.posterior(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    return lp, pm, pv


def _check(tgp, dm, yin, Rn, lp, pm, pv, kernel, served=True):
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    got = tgp.logpdf(dm, yin)
    info = hd.sweep_info()
    assert info["served"] == int(served), info
    assert abs(got - lp) <= 1e-10 * abs(lp), (got, lp, info)
    got2, mean, var = tgp.logpdf_and_posterior_marginals(dm, yin, Rn)
    info = hd.sweep_info()
    assert info["served"] == int(served), info
    assert abs(got2 - lp) <= 1e-10 * abs(lp), (got2, lp, info)
    assert np.abs(mean - pm).max() <= 1e-8 * max(1.0, np.abs(pm).max()), info
    assert np.abs(var - pv).max() <= 1e-8 * max(1.0, pv.max()), info
    names = set(hd.profile())
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    if served:
        assert names == {f"k_sweep<{kernel},logpdf>", f"k_sweep<{kernel},posterior>"}, names
    else:
        assert not any(n.startswith("k_sweep") for n in names) or info["attempts"] >= 1
    return info


@pytest.mark.parametrize("i", range(len(KERNELS)))
@pytest.mark.parametrize("T", [2048, 5003, 40_000])
def test_missing_data_on_a_regular_grid(tgp, i, T):
    k, dt, s2 = KERNELS[i]
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=i)
    rng = np.random.default_rng(100 + i)
    missing = rng.random(T) < 0.1
    missing[5:9] = True
    missing[T - 3:] = True
    lp, pm, pv = _reference(model, y, missing, 1e-18, fast=T > 6000)
    yin = np.where(missing, np.nan, y)
    info = _check(tgp, _lti_device_model(tgp, model), yin, np.array([1e-18]), lp, pm, pv, "lti")
    assert info["attempts"] == 1 and info["Wb"] <= info["C"]


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_per_step_noise_offset_and_new_noise(tgp, i):
    k, dt, s2 = KERNELS[i]
    T = 4500
    rng = np.random.default_rng(7 + i)
    S = s2 * (0.5 + rng.random(T))
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), S, seed=i, mean=("custom", lambda t: np.sin(t)))
    Rn = rng.random(T) * 0.05
    lp, pm, pv = _reference(model, y, None, Rn)
    _check(tgp, _lti_device_model(tgp, model), y, Rn, lp, pm, pv, "lti")
    # ... and with missing steps on top
    missing = rng.random(T) < 0.2
    lp, pm, pv = _reference(model, y, missing, Rn)
    _check(tgp, _lti_device_model(tgp, model), np.where(missing, np.nan, y), Rn, lp, pm, pv, "lti")


@pytest.mark.parametrize("i", range(len(KERNELS)))
@pytest.mark.parametrize("with_missing", [False, True])
def test_irregular_spacing(tgp, i, with_missing):
    """broadcast_components for AbstractVector inputs (lti_sde.jl:135-146) through build_lgssm's device-side transitions"""
    from temporalgps_jl_amd import lti_sde as P
    k, dt, s2 = KERNELS[i]
    T = 6000
    rng = np.random.default_rng(40 + i)
    t = np.cumsum(rng.uniform(0.5 * dt, 1.5 * dt, T))
    model, y, _ = U.gp_case(k, t, s2, seed=i)
    missing = (rng.random(T) < 0.15) if with_missing else None
    lp, pm, pv = _reference(model, y, missing, 1e-18)
    dm = P.build_lgssm(P.to_kernel(k), t, s2, device_components=True)
    yin = y if missing is None else np.where(missing, np.nan, y)
    _check(tgp, dm, yin, np.array([1e-18]), lp, pm, pv, "sde")


@pytest.mark.parametrize("i", [2, 3, 4])
def test_irregular_spacing_with_per_step_noise_and_offset(tgp, i):
    """every input stream at once (gaps, noise variance, emission offset of a mean function at the inputs, mask, new noise per step): the
    kernel variant with the largest register footprint (tests/test_kernel_resources.py names this test for it), d = 3 and 4"""
    from temporalgps_jl_amd import lti_sde as P
    k, dt, s2 = KERNELS[i]
    T = 5000
    rng = np.random.default_rng(60 + i)
    t = np.cumsum(rng.uniform(0.5 * dt, 1.5 * dt, T))
    S = s2 * (0.5 + rng.random(T))
    model, y, _ = U.gp_case(k, t, S, seed=i, mean=("custom", lambda tt: np.cos(0.3 * tt)))
    missing = rng.random(T) < 0.1
    Rn = rng.random(T) * 0.05
    lp, pm, pv = _reference(model, y, missing, Rn)
    dm = P.build_lgssm(P.to_kernel(k), t, S, mean=P.CustomMean(lambda v: np.cos(0.3 * v)), device_components=True)
    _check(tgp, dm, np.where(missing, np.nan, y), Rn, lp, pm, pv, "sde")


def test_prediction_at_new_inputs_runs_the_sweep_engine(tgp):
    """the reference's predict path (posterior_lti_sde.jl:20-37,97-131): training and prediction inputs merged and sorted, the prediction
    points missing with the large noise variance of missings.jl:43"""
    from temporalgps_jl_amd import lti_sde as P
    rng = np.random.default_rng(5)
    k = ("scaled", 0.8, ("stretched", 1.7, ("matern52",)))
    ntr, npr = 3000, 1500
    xtr = np.sort(rng.uniform(0.0, 300.0, ntr))
    xpr = np.sort(rng.uniform(-1.0, 301.0, npr))
    model_tr, ytr, _ = U.gp_case(k, xtr, 0.1, seed=9)
    pm, pv = oc.posterior_marginals(k, xtr, 0.1, ytr, x_pr=xpr, sigma2_pr=1e-18)
    f = P.to_sde(P.GP(P.to_kernel(k)))
    post = P.posterior(f(xtr, 0.1), ytr)
    mean, var = P.mean_and_var(post(xpr))
    assert np.abs(mean - pm).max() <= 1e-8 * max(1.0, np.abs(pm).max())
    assert np.abs(var - pv).max() <= 1e-8 * max(1.0, pv.max())


def test_warm_ups_that_prove_too_short_are_repeated_longer(tgp):
    """the bench parametrisation (dt = 0.05 in stretched time, l = 2.3) mixes slowly: start the engine with warm-ups that are too short by
    construction (the hint a previous call left) -- the checks must catch it and the call must still return the right numbers"""
    k, dt, s2 = KERNELS[5]
    T = 30_000
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=3)
    missing = np.random.default_rng(1).random(T) < 0.1
    lp, pm, pv = _reference(model, y, missing, 1e-18, fast=True)
    dm = _lti_device_model(tgp, model)
    hd = dm.handle()
    yin = np.where(missing, np.nan, y)
    # forced geometry: never repaired -> the general engine serves the call, correctly
    hd.set_option(tgp._lib.OPT_SWEEP_WARMUP, 16)
    hd.set_option(tgp._lib.OPT_SWEEP_WARMUP_BACK, 16)
    got = tgp.logpdf(dm, yin)
    info = hd.sweep_info()
    assert info["served"] == 0 and info["status"] & 1 and info["attempts"] == 1, info
    assert abs(got - lp) <= 1e-10 * abs(lp)
    _, mean, var = tgp.logpdf_and_posterior_marginals(dm, yin, np.array([1e-18]))
    info = hd.sweep_info()
    assert info["served"] == 0 and info["status"] & 3, info
    assert np.abs(mean - pm).max() <= 1e-8 * max(1.0, np.abs(pm).max())
    hd.set_option(tgp._lib.OPT_SWEEP_WARMUP, 0)
    hd.set_option(tgp._lib.OPT_SWEEP_WARMUP_BACK, 0)
    got, mean, var = tgp.logpdf_and_posterior_marginals(dm, yin, np.array([1e-18]))
    info = hd.sweep_info()
    assert info["served"] == 1, info
    assert abs(got - lp) <= 1e-10 * abs(lp)
    assert np.abs(mean - pm).max() <= 1e-8 * max(1.0, np.abs(pm).max())
    assert np.abs(var - pv).max() <= 1e-8 * max(1.0, pv.max())


def test_device_resident_inputs_and_outputs_at_a_million_steps(tgp):
    import torch
    k, dt, s2 = KERNELS[2]
    T = 1_000_000
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, 64), s2, seed=0)
    model = dict(model, T=T)
    rng = np.random.default_rng(17)
    y = rng.standard_normal(T)
    missing = rng.random(T) < 0.1
    lp, pm, pv = _reference(model, y, missing, 1e-18, fast=True)
    dm = _lti_device_model(tgp, model)
    yd = torch.as_tensor(np.where(missing, 0.0, y), device="cuda:0")
    md = torch.as_tensor(missing, device="cuda:0")
    rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    got, mean, var = tgp.logpdf_and_posterior_marginals(dm, (yd, md), rn)
    info = dm.handle().sweep_info()
    assert info["served"] == 1 and info["attempts"] == 1, info
    assert abs(got - lp) <= 1e-10 * abs(lp)
    assert float((mean.cpu() - torch.as_tensor(pm)).abs().max()) <= 1e-8 * max(1.0, np.abs(pm).max())
    assert float((var.cpu() - torch.as_tensor(pv)).abs().max()) <= 1e-8


def test_the_engine_can_be_switched_off_and_leaves_the_other_engines_alone(tgp):
    k, dt, s2 = KERNELS[2]
    T = 5000
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=2)
    missing = np.random.default_rng(3).random(T) < 0.1
    lp, pm, pv = _reference(model, y, missing, 1e-18)
    dm = _lti_device_model(tgp, model)
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_SWEEP, 0)
    got = tgp.logpdf(dm, np.where(missing, np.nan, y))
    assert hd.sweep_info()["served"] == 0 and abs(got - lp) <= 1e-10 * abs(lp)
    # a fully observed LTI series with one noise variance is the stationary-gain engines' (the sweep engine is never asked)
    hd.set_option(tgp._lib.OPT_SWEEP, 1)
    lp2 = ref.logpdf(model, y)
    got = tgp.logpdf(dm, y)
    assert hd.sweep_info()["served"] == 0 and abs(got - lp2) <= 1e-10 * abs(lp2)
