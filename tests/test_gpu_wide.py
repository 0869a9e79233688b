"""GPU tier: the stationary-gain engine for WIDE states (8 < d <= 63; csrc/tgp_wide.hip, round 6) -- logpdf of Forward LTI models with scalar observations
on the dense closed loop, one wave per chunk -- against the literal restatement of lgssm.jl:147-165 (oracle/lgssm_ref.py) at lengths its Python loops
finish, and against the dense engine's sequential passes (TGP_OPT_WIDE = 0) beyond.  Products of kernels (lti_sde.jl:377-400) are what produces such states:
ApproxPeriodicKernel() * Matern32Kernel() has d = 28.  Tolerance as everywhere: 1e-10 relative."""
import os

import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref

pytestmark = pytest.mark.gpu

# (TGP_WIDE_DPP=0 in the environment: the LDS kernels for every d -- the A/B run of DESIGN 4.4)
NARROW = "k_wide_lml<32>" if os.environ.get("TGP_WIDE_DPP") == "0" else "k_wide_lml4"
KERNELS = {
    9: ("product", ("matern52",), ("stretched", 0.7, ("matern52",))),
    12: ("product", ("matern32",), ("approx_periodic", 3, 1.0)),
    15: ("sum", ("matern52",), ("stretched", 0.5, ("matern52",)), ("stretched", 2.0, ("matern52",)), ("stretched", 0.3, ("matern52",)), ("scaled", 0.5, ("stretched", 1.5, ("matern52",)))),
    16: ("product", ("approx_periodic", 4, 1.0), ("matern32",)),
    18: ("product", ("approx_periodic", 3, 1.0), ("matern52",)),
    28: ("product", ("approx_periodic", 7, 1.0), ("matern32",)),
    42: ("product", ("approx_periodic", 7, 1.0), ("matern52",)),
}


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def device_model(tgp, model, wide=1):
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])
    dm.handle_options[tgp._lib.OPT_WIDE] = wide
    return dm


def kernels_of(tgp, dm, fn):
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    out = fn()
    names = set(hd.profile())
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    return out, names


def draw(model, seed):
    T, d = model["T"], len(model["x0m"])
    rng = np.random.default_rng(seed)
    return ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))


@pytest.mark.parametrize("d", sorted(KERNELS))
def test_wide_logpdf_against_the_restatement(tgp, d):
    for T, dt, s2 in ((6000, 0.1, 0.1), (20_000, 0.05, 0.02)) if d == 28 else ((2500, 0.1, 0.1),) if d % 2 else ((3000, 0.05, 0.02),):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), s2)
        assert len(model["x0m"]) == d
        y = draw(model, d + T)
        lp_ref = ref.logpdf(model, y)
        dm = device_model(tgp, model)
        lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (d, T, lp, lp_ref)
        assert names == {(NARROW + "<16>" if d <= 15 and NARROW == "k_wide_lml4" else NARROW) if d <= 31 else ("k_wide_lml4<48>" if d <= 47 and NARROW == "k_wide_lml4" else "k_wide_lml<64>")}, names
        # a second call of the same model keeps the plan; another series, the same answer as the dense engine's sequential pass
        y2 = draw(model, d + T + 1)
        lp2 = tgp.logpdf(dm, y2)
        if d > 16:      # (below: the general engine's code object of that d is 20 s to load; scripts/stress_wide.py compares with it)
            dm0 = device_model(tgp, model, wide=0)
            lp2_dense = tgp.logpdf(dm0, y2)
            assert abs(lp2 - lp2_dense) <= 1e-10 * abs(lp2_dense), (d, T, lp2, lp2_dense)
        else:
            lp2_ref = ref.logpdf(model, y2)
            assert abs(lp2 - lp2_ref) <= 1e-10 * abs(lp2_ref), (d, T, lp2, lp2_ref)


def test_wide_logpdf_long_series_device_input(tgp):
    import torch
    T = 400_000
    model = oc.build_lgssm(KERNELS[28], ("regular", 0.0, 0.1, T), 0.1)
    rng = np.random.default_rng(3)
    # (a draw of the model's own scale without the restatement's Python loop: white noise of the prior's marginal variance is as good a series for parity)
    y = rng.standard_normal(T) * np.sqrt(float(model["H"][0] @ model["x0P"] @ model["H"][0]) + 0.1)
    yd = torch.from_numpy(y).cuda()
    dm, dm0 = device_model(tgp, model), device_model(tgp, model, wide=0)
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, yd))
    assert names == {NARROW}, names
    lp0, names0 = kernels_of(tgp, dm0, lambda: tgp.logpdf(dm0, yd))
    assert not any(n.startswith("k_wide") for n in names0), names0
    assert abs(lp - lp0) <= 1e-10 * abs(lp0), (lp, lp0)


def test_series_too_short_for_the_engine_goes_to_the_dense_passes(tgp):
    T = 120      # (the covariance settles after 87 steps: 33 steps behind the head are fewer than one chunk)
    model = oc.build_lgssm(KERNELS[28], ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 1)
    dm = device_model(tgp, model)
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    lp_ref = ref.logpdf(model, y)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    assert not any(n.startswith("k_wide") for n in names), names


def dense_gp_posterior(model, y, Rn):
    """The posterior marginals from the model's OWN covariance function, k(s - t) = h' A^|s - t| P_inf h (x0 stationary, no offsets: what to_sde builds), by a
    dense Cholesky -- no state-space recursion involved, conditioned like K + R I (1e3), not like the predicted covariances."""
    from scipy.linalg import cho_factor, cho_solve, toeplitz
    T = model["T"]
    A, H, P, R = model["A"][0], model["H"][0], model["x0P"], float(model["R"][0])
    c, v = np.empty(T), P @ H
    for k in range(T):
        c[k] = H @ v
        v = A @ v
    K = toeplitz(c)
    cf = cho_factor(K + R * np.eye(T), lower=True)
    return K @ cho_solve(cf, y), np.diag(K) - np.einsum("ij,ji->i", K, cho_solve(cf, K)) + Rn


@pytest.mark.parametrize("d", (9, 15, 16, 28, 42))      # one / two components per lane with the observer in either half, the LDS kernels
def test_wide_posterior_marginals(tgp, d):
    """marginals(replace_observation_noise_cov(posterior(model, y), Rnew)) (lgssm.jl:99-115, 193-238): forward kernel keeping its innovations, backward
    kernel in Bryson-Frazier form, the head and the last n1 steps' variances from the host's tables.  Two references: the dense GP on the model's own
    covariance function at 1e-8 (independent of every recursion), and the literal RTS restatement at 1e-6 -- invert_dynamics solves with the predicted
    covariance, whose condition at d = 28 costs the restatement itself 5e-8 of the mean against the dense GP (the Bryson-Frazier form has no solve)"""
    rng = np.random.default_rng(d)
    for T, dt, s2 in ((2500, 0.1, 0.1), (4000, 0.05, 0.02)) if d == 28 else ((2500, 0.1, 0.1),) if d % 2 else ((3000, 0.05, 0.02),):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), s2)
        y = draw(model, 2 * d + T)
        lp_ref = ref.logpdf(model, y)
        post = ref.posterior(model, y)
        for per_step in (False, True):
            Rn = rng.random(T) * 0.3 + 0.01 if per_step else np.array([0.05])
            Rfull = Rn if per_step else np.full(T, Rn[0])
            m_rts, v_rts = ref.marginals(ref.replace_observation_noise_cov(post, Rfull))
            m_rts, v_rts = np.asarray(m_rts).reshape(T), np.asarray(v_rts).reshape(T)
            m_gp, v_gp = dense_gp_posterior(model, y, Rfull)
            dm = device_model(tgp, model)
            (lp, mean, var), names = kernels_of(tgp, dm, lambda: tgp.logpdf_and_posterior_marginals(dm, y, Rn))
            assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (d, T, lp, lp_ref)
            assert np.max(np.abs(mean - m_gp)) <= 1e-8 * max(1.0, np.abs(m_gp).max()), (d, T, per_step, np.max(np.abs(mean - m_gp)))
            assert np.max(np.abs(var - v_gp)) <= 1e-8 * max(1.0, v_gp.max()), (d, T, per_step, np.max(np.abs(var - v_gp)))
            assert np.max(np.abs(mean - m_rts)) <= 1e-6 * max(1.0, np.abs(m_rts).max()) and np.max(np.abs(var - v_rts)) <= 1e-6 * max(1.0, v_rts.max())
            assert len(names) == 1 and next(iter(names)).startswith("k_wide_lml"), names


def test_wide_posterior_long_series_device_arrays(tgp):
    import torch
    T = 60_000      # (the dense engine's RTS chain beside it takes 80 us per step)
    model = oc.build_lgssm(KERNELS[28], ("regular", 0.0, 0.1, T), 0.1)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(T) * np.sqrt(float(model["H"][0] @ model["x0P"] @ model["H"][0]) + 0.1)
    yd, Rn = torch.from_numpy(y).cuda(), torch.full((1,), 0.2, dtype=torch.float64, device="cuda")
    dm, dm0 = device_model(tgp, model), device_model(tgp, model, wide=0)
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, yd, Rn))
    assert next(iter(names)).startswith("k_wide_lml"), names
    # (the dense engine runs the reference's RTS chain with its solves against the predicted covariance: 1e-6, as in test_wide_posterior_marginals)
    mean0, var0 = tgp.posterior_marginals(dm0, yd, Rn)
    assert float((mean - mean0).abs().max()) <= 1e-6 * max(1.0, float(mean0.abs().max()))
    assert float((var - var0).abs().max()) <= 1e-6 * max(1.0, float(var0.max()))


@pytest.mark.parametrize("d", (12, 28))
def test_wide_prior_marginals_settle_on_the_host(tgp, d):
    """marginals(model) of an LTI prior (lgssm.jl:99-109): the (m, P) recursion on the host until it no longer moves, one fill kernel -- for wide states too
    (before: the general engine's affine scan up to d = 16, the dense engine's sequential pass beyond)"""
    rng = np.random.default_rng(d)
    T = 3000
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    # (a prior that does not start in its stationary state: the recursion has a transient to follow)
    U = np.linalg.qr(rng.standard_normal((d, d)))[0]
    model["x0m"] = rng.standard_normal(d)
    model["x0P"] = (U * (rng.random(d) + 0.5)) @ U.T
    m_ref, v_ref = ref.marginals(model)
    m_ref, v_ref = np.asarray(m_ref).reshape(T), np.asarray(v_ref).reshape(T)
    dm = device_model(tgp, model)
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.marginals(dm))
    assert np.max(np.abs(mean - m_ref)) <= 1e-10 * max(1.0, np.abs(m_ref).max()) and np.max(np.abs(var - v_ref)) <= 1e-10 * max(1.0, v_ref.max())
    assert names == {"k_fill_marginals<lti>"}, names


@pytest.mark.parametrize("d", (12, 28, 42))
def test_wide_model_with_a_mean_function_at_the_inputs(tgp, d):
    """an emission offset PER STEP (a GP with a mean function on a regular grid: everything else shared): the gains do not see it -- the kernels subtract
    h_t where they load y_t.  logpdf against the restatement, posterior marginals against the dense GP on y - h (mean + h), prior marginals on the
    engine of before (the host recursion does not serve a per-step offset)"""
    rng = np.random.default_rng(100 + d)
    T = 2500
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    ht = 0.8 * np.sin(0.013 * np.arange(T)) + 0.0004 * np.arange(T) - 0.5
    y = draw(model, d) + ht
    model_h = dict(model, h=ht)
    lp_ref = ref.logpdf(model_h, y)
    for per_step in (False, True):
        Rn = rng.random(T) * 0.3 + 0.01 if per_step else np.array([0.05])
        m_gp, v_gp = dense_gp_posterior(model, y - ht, Rn if per_step else np.full(T, Rn[0]))
        dm = device_model(tgp, model_h)
        (lp, mean, var), names = kernels_of(tgp, dm, lambda: tgp.logpdf_and_posterior_marginals(dm, y, Rn))
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (d, lp, lp_ref)
        assert np.max(np.abs(mean - (m_gp + ht))) <= 1e-8 * max(1.0, np.abs(m_gp).max()) and np.max(np.abs(var - v_gp)) <= 1e-8 * max(1.0, v_gp.max())
        assert len(names) == 1 and next(iter(names)).startswith("k_wide_lml"), names
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref) and next(iter(names)).startswith("k_wide_lml"), names
    m_ref, v_ref = ref.marginals(model_h)
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.marginals(dm))
    assert np.max(np.abs(mean - np.asarray(m_ref).reshape(T))) <= 1e-9 and np.max(np.abs(var - np.asarray(v_ref).reshape(T))) <= 1e-9
    assert "k_fill_marginals<lti>" not in names, names


def test_gp_level_api_on_a_product_kernel(tgp, monkeypatch):
    """The reference's own call chain (lti_sde.jl:33-68, posterior_lti_sde.jl:18-37) on ApproxPeriodicKernel() * Matern32Kernel(): logpdf(fx, y) and
    marginals(posterior(fx, y)(x)) bind ONE model whose calls run on the wide engine; values against the dense GP on the kernel itself."""
    from oracle import dense_gp as dg
    from temporalgps_jl_amd import lti_sde as P
    rng = np.random.default_rng(2)
    T = 1800
    spec = KERNELS[28]
    x = P.RegularSpacing(0.0, 0.1, T)
    f = P.to_sde(P.GP(P.ApproxPeriodicKernel() * P.Matern32Kernel()), P.HIPStorage())
    fx = f(x, 0.1)
    y = np.asarray(P.rand(rng, fx))
    built, real = [], P.build_lgssm

    def profiled(*a, **k):      # (every model the API binds from here on records its kernels)
        mdl = real(*a, **k)
        mdl.handle_options[tgp._lib.OPT_PROFILE] = 1
        built.append(mdl)
        return mdl
    monkeypatch.setattr(P, "build_lgssm", profiled)
    lp = P.logpdf(fx, y)
    m, sd = P.marginals(P.posterior(fx, y)(x, 1e-9))
    m2, sd2 = P.marginals(P.posterior(fx, y)(x, 1e-9))
    assert built and all(b.dim == 28 and b.T == T for b in built)
    names = set()
    for b in built:
        if b._handle is not None:
            names |= set(b.handle().profile())
    assert names and all(n.startswith("k_wide_lml") for n in names), names
    xs = x.collect()
    lp_d = dg.logpdf(spec, xs, 0.1, y)
    md, vd = dg.posterior_marginals(spec, xs, 0.1, y, xs, 1e-9)
    # (ApproxPeriodicKernel{7} approximates the periodic kernel the dense GP is built on to ~1e-7: test/gp/lti_sde.jl:113-116, tests/test_gpu_parity.py)
    assert abs(lp - lp_d) <= 1e-5 * abs(lp_d), (lp, lp_d)
    assert np.max(np.abs(np.asarray(m) - md)) <= 1e-4 and np.max(np.abs(np.asarray(sd) ** 2 - vd)) <= 1e-4
    np.testing.assert_array_equal(np.asarray(m), np.asarray(m2))


@pytest.mark.parametrize("d", (12, 28, 42))
def test_wide_rand_with_the_draws_supplied(tgp, d):
    """rand(model) (lgssm.jl:65-91) of a wide LTI prior in ONE kernel on the open loop (k_wide_rand: chunks warmed up on the same draws) against the literal
    restatement on the same draws; a long series against the engine of before"""
    import torch
    rng = np.random.default_rng(7 * d)
    T = 3000
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y_ref = ref.rand(model, *eps)
    dm = device_model(tgp, model)
    y, names = kernels_of(tgp, dm, lambda: tgp.rand(eps, dm))
    assert np.max(np.abs(y - y_ref)) <= 1e-9 * max(1.0, np.abs(y_ref).max()), (d, np.max(np.abs(y - y_ref)))
    assert names == {"k_wide_rand"}, names
    if d == 28:
        T2 = 200_000
        model2 = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T2), 0.1)
        eps2 = (torch.randn((T2, d), dtype=torch.float64, device="cuda"), torch.randn((T2,), dtype=torch.float64, device="cuda"), rng.standard_normal(d))
        y1 = tgp.rand(eps2, device_model(tgp, model2))
        y0 = tgp.rand(eps2, device_model(tgp, model2, wide=0))
        assert float((y1 - y0).abs().max()) <= 1e-9 * max(1.0, float(y0.abs().max()))


@pytest.mark.parametrize("d", (9, 15, 28, 42))
def test_wide_filter(tgp, d):
    """_filter (lgssm.jl:171-187): filtered means from the forward kernel, the head's covariances from the host plan, the settled covariance behind them"""
    import torch
    T = 2500
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 11 * d)
    fm_ref, fP_ref = ref.filter_(model, y)
    dm = device_model(tgp, model)
    for yy in (y, torch.from_numpy(y).cuda()):
        (fm, fP), names = kernels_of(tgp, dm, lambda: tgp._filter(dm, yy))
        fm, fP = (fm.cpu().numpy(), fP.cpu().numpy()) if hasattr(fm, "cpu") else (fm, fP)
        assert np.max(np.abs(fm - fm_ref)) <= 1e-8 * max(1.0, np.abs(fm_ref).max()), (d, np.max(np.abs(fm - fm_ref)))
        assert np.max(np.abs(fP - fP_ref)) <= 1e-8 * max(1.0, np.abs(fP_ref).max()), (d, np.max(np.abs(fP - fP_ref)))
        assert len(names) == 1 and next(iter(names)).startswith("k_wide_lml"), names
