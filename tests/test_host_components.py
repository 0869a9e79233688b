"""CPU tier: the product's host-side component construction (temporalgps.jl_amd/lti_sde.py, the mirror of
/root/reference/src/gp/lti_sde.jl:112-445) against the oracle's independent restatement and the dense GP."""
import numpy as np
import pytest

import temporalgps_jl_amd  # noqa: F401  (registers the package)
from temporalgps_jl_amd import lti_sde as P
from oracle import components as oc
from oracle import dense_gp as dg
from tests.test_oracle_identities import KERNELS, N


@pytest.mark.parametrize("kname", list(KERNELS) + ["approx-periodic"])
@pytest.mark.parametrize("spacing", ["regular", "irregular"])
def test_components_match_oracle(kname, spacing):
    spec = ("approx_periodic", 7, 1.0) if kname == "approx-periodic" else KERNELS[kname]
    t_o = ("regular", 0.0, 0.3, N) if spacing == "regular" else 0.3 * np.arange(N)
    t_p = P.RegularSpacing(0.0, 0.3, N) if spacing == "regular" else 0.3 * np.arange(N)
    A, a, Q, H, h, (m0, P0) = P.to_kernel(spec).lgssm_components(t_p)
    Ao, ao, Qo, Ho, ho, (m0o, P0o) = oc.lgssm_components(spec, t_o)
    for got, want in ((A, Ao), (Q, Qo), (H, Ho), (h, ho), (m0, m0o), (P0, P0o)):
        want = np.asarray(want)
        got = np.asarray(got)
        if got.shape != want.shape:
            got, want = np.broadcast_arrays(got, want)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)
    assert (A.shape[0] == 1) == (spacing == "regular")          # Fill <=> regular spacing (lti_sde.jl:148-160)


def test_kernel_algebra_operators():
    k = 1.5 * P.Matern12Kernel().stretch(0.1) + 0.3 * P.Matern32Kernel().stretch(1.1)
    t = P.RegularSpacing(0.0, 0.3, N)
    A, a, Q, H, h, x0 = k.lgssm_components(t)
    Ao, ao, Qo, Ho, ho, x0o = oc.lgssm_components(KERNELS["sum-12-32"], ("regular", 0.0, 0.3, N))
    np.testing.assert_allclose(A, Ao, rtol=1e-13)
    np.testing.assert_allclose(H, Ho, rtol=1e-13)
    prod = 3.0 * P.Matern32Kernel() * P.Matern52Kernel() * P.ConstantKernel()
    assert isinstance(prod, P.ScaledKernel) or isinstance(prod, P.KernelProduct)


def test_mean_functions_and_noise_shapes():
    t = P.RegularSpacing(0.0, 0.3, N)
    m = P.build_lgssm(P.Matern32Kernel(), t, 0.1, P.ConstMean(3.0))
    assert np.allclose(m.emissions.h, 3.0)              # a constant mean keeps the Fill layout
    m = P.build_lgssm(P.Matern32Kernel(), t, 0.1, P.CustomMean(lambda x: 2 * x))
    np.testing.assert_allclose(m.emissions.h, 2 * t.collect())
    m = P.build_lgssm(P.Matern32Kernel(), t, 0.1, force_per_step=True)
    assert m.transitions.As.shape == (N, 2, 2) and m.emissions.R.shape == (N,)
    with pytest.raises(ValueError):
        P.to_sde(P.GP(P.Matern32Kernel()))(t, np.ones(N + 1))
    with pytest.raises(ValueError):
        P.HIPStorage(np.float32)


def test_regular_spacing_matches_range():
    # test/util/regular_data.jl:8-12
    x = P.RegularSpacing(0.1, 0.25, 17)
    np.testing.assert_allclose(x.collect(), 0.1 + 0.25 * np.arange(17))
    assert len(x) == 17 and x[3] == 0.1 + 3 * 0.25 and x.step() == 0.25


def test_posterior_merge_bookkeeping():
    """merge_datasets (posterior_lti_sde.jl:97-123): indices recover both data sets from the sorted union."""
    rng = np.random.default_rng(0)
    f = P.to_sde(P.GP(P.Matern32Kernel()))
    x_tr, x_pr = np.sort(rng.random(9)), np.sort(rng.random(4)) + 0.001
    y_tr = rng.standard_normal(9)
    fpost = P.posterior(f(x_tr, 0.1), y_tr)(x_pr, 0.2)
    x, S, y, tr, pr = fpost._merge(np.full(4, fpost.LARGE_VAR))
    assert np.all(np.diff(x) >= 0)
    np.testing.assert_array_equal(x[tr], x_tr)
    np.testing.assert_array_equal(x[pr], x_pr)
    np.testing.assert_array_equal(y[tr], y_tr)
    assert np.all(np.isnan(y[pr])) and np.all(S[pr] == 1e15) and np.all(S[tr] == 0.1)


@pytest.mark.parametrize("noise", ["scalars", "per-step", "missing"])
def test_pair_statistic_gives_the_posterior_logpdf_at_the_training_inputs(noise):
    """`_pair_statistic` (the same-inputs route of posterior_lti_sde.jl:62-78 in the mirror): with two observations per input,
    log p(y* | y) = log N(ybar; m, K + Rbar) + const - log N(y; m, K + R) -- held against the dense GP's posterior logpdf."""
    rng = np.random.default_rng(3)
    n = 25
    spec = KERNELS["sum-12-32"]
    x = np.sort(rng.random(n)) * 4
    y, ys = rng.standard_normal(n), rng.standard_normal(n)
    R, Rs = (np.array([0.3]), np.array([0.07])) if noise == "scalars" else (rng.random(n) * 0.3 + 0.05, rng.random(n) * 0.2 + 0.01)
    ym, ysm = y.copy(), ys.copy()
    if noise == "missing":
        ym[[2, 9, 11]] = np.nan
        ysm[[4, 9, 20]] = np.nan
    ybar, Rbar, const = P._pair_statistic(ym, R, ysm, Rs)
    assert (Rbar.shape == (1,)) == (noise == "scalars")
    kt, kp, kj = ~np.isnan(ym), ~np.isnan(ysm), ~np.isnan(ybar)
    Rf, Rsf, Rbf = np.broadcast_to(R, (n,)), np.broadcast_to(Rs, (n,)), np.broadcast_to(Rbar, (n,))
    got = dg.logpdf(spec, x[kj], Rbf[kj], ybar[kj]) + const - dg.logpdf(spec, x[kt], Rf[kt], y[kt])
    want = dg.posterior_logpdf(spec, x[kt], Rf[kt], y[kt], x[kp], Rsf[kp], ys[kp])
    assert abs(got - want) <= 1e-9 * abs(want)


@pytest.mark.parametrize("noise", ["scalars", "per-step"])
@pytest.mark.parametrize("gaps", [False, True])
def test_posterior_logpdf_pair_host_logic_against_the_oracles_literal_chain(monkeypatch, noise, gaps):
    """lgssm.py `_posterior_logpdf_pair` (host arrays) with the device calls replaced by the oracle's logpdf: the value of
    logpdf(replace_observation_noise_cov(posterior(model, y), R_new), y_new) against the oracle's literal chain -- posterior evaluated
    (lgssm.jl:193-221), noise replaced (missings.jl:35-41), filtered (lgssm.jl:147-151) -- with gaps on either side and on both."""
    from oracle import lgssm_ref as ref
    from temporalgps_jl_amd import lgssm as L
    rng = np.random.default_rng(12)
    T = 300
    s2 = 0.3 if noise == "scalars" else rng.random(T) * 0.3 + 0.1
    model = oc.build_lgssm(KERNELS["sum-12-32"], ("regular", 0.0, 0.2, T), s2)
    d = len(model["x0m"])
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y_new = y + 0.4 * rng.standard_normal(T)
    R_new = np.array([0.2]) if noise == "scalars" else rng.random(T) * 0.2 + 0.05
    miss, miss_new = np.zeros(T, dtype=bool), np.zeros(T, dtype=bool)
    if gaps:
        miss[[3, 100, 101]] = True
        miss_new[[7, 100, 299]] = True

    def as_oracle(m):
        em = m.emissions
        return dict(model, R=np.asarray(em.R, dtype=np.float64).reshape(-1))

    def oracle_logpdf(m, yy):
        yy = np.asarray(yy, dtype=np.float64)
        return ref.logpdf_missing(as_oracle(m), np.nan_to_num(yy), np.isnan(yy))

    monkeypatch.setattr(L, "logpdf", oracle_logpdf)
    monkeypatch.setattr(L, "_logpdf_with_noise", lambda prior, ybar, Rbar: None)
    tr = L.GaussMarkovModel(L.Forward, model["A"], model["a"], model["Q"], L.Gaussian(model["x0m"], model["x0P"]))
    prior = L.LGSSM(tr, L.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
    ym, ynm = np.where(miss, np.nan, y), np.where(miss_new, np.nan, y_new)
    got = L._posterior_logpdf_pair(L.replace_observation_noise_cov(L.posterior(prior, ym), R_new), ynm)
    post = ref.replace_observation_noise_cov(ref.posterior_missing(model, y, miss), R_new)
    want = ref.logpdf_missing(post, y_new, miss_new)
    assert got is not None and abs(got - want) <= 1e-8 * abs(want)
