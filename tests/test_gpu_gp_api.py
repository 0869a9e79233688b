"""GPU tier: the GP-level API surface (to_sde / FiniteGP methods / posterior queries), i.e. the callers of the hot path
(/root/reference/src/gp/lti_sde.jl:33-68, src/gp/posterior_lti_sde.jl:18-78), through the device engine, against the
oracle's restatement of the same functions and against the dense GP (the reference's bar: rtol 1e-5 for posteriors at
new inputs, test/gp/posterior_lti_sde.jl:82-89)."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import dense_gp as dg
from tests.test_oracle_identities import KERNELS

pytestmark = pytest.mark.gpu
NAMES = ["base-Matern12", "base-Matern32", "base-Matern52", "scaled-10.0", "stretched-0.1", "sum-12-32", "prod-52-32"]


@pytest.fixture(scope="module")
def P():
    import temporalgps_jl_amd  # noqa: F401
    from temporalgps_jl_amd import lti_sde
    return lti_sde


@pytest.mark.parametrize("kname", NAMES)
@pytest.mark.parametrize("spacing", ["regular", "irregular"])
def test_prior_api_equals_dense_gp(P, kname, spacing):
    rng = np.random.default_rng(3)
    spec = KERNELS[kname]
    N = 40
    x = P.RegularSpacing(0.0, 0.3, N) if spacing == "regular" else np.cumsum(rng.random(N) * 0.4 + 0.05)
    xs = x.collect() if spacing == "regular" else x
    s2 = rng.random(N) * 0.2 + 0.1
    fx = P.to_sde(P.GP(P.CustomMean(lambda t: 0.5 * t), P.to_kernel(spec)), P.HIPStorage())(x, s2)
    y = P.rand(rng, fx)
    assert y.shape == (N,)
    mean = ("custom", lambda t: 0.5 * t)
    lp, lp_d = P.logpdf(fx, y), dg.logpdf(spec, xs, s2, y, mean)
    assert abs(lp - lp_d) <= 1.5e-8 * abs(lp_d) + 1e-9
    m, sd = P.marginals(fx)
    md, vd = dg.marginals(spec, xs, s2, mean)
    np.testing.assert_allclose(m, md, rtol=1.5e-8, atol=1e-10)
    np.testing.assert_allclose(sd ** 2, vd, rtol=1.5e-8)
    mv = P.mean_and_var(fx)
    np.testing.assert_allclose(mv[0], P.mean(fx))
    np.testing.assert_allclose(mv[1], P.var(fx))
    ym = y.copy()
    ym[[3, 17]] = np.nan                                       # missing observations (missings.jl)
    keep = ~np.isnan(ym)
    assert abs(P.logpdf(fx, ym) - dg.logpdf(spec, xs[keep], s2[keep], y[keep], mean)) <= 1.5e-8 * abs(lp_d) + 1e-8


@pytest.mark.parametrize("kname", ["base-Matern32", "base-Matern52", "sum-12-32"])
def test_posterior_api_same_and_new_inputs(P, kname):
    rng = np.random.default_rng(7)
    spec = KERNELS[kname]
    x_tr = np.sort(rng.random(30)) * 5
    s_tr = rng.random(30) * 0.2 + 0.05
    y_tr = rng.standard_normal(30)
    f = P.to_sde(P.GP(P.to_kernel(spec)), P.HIPStorage())
    fpost = P.posterior(f(x_tr, s_tr), y_tr)
    # same inputs (posterior_lti_sde.jl:27-36)
    m, sd = P.marginals(fpost(x_tr, 0.3))
    md, vd = dg.posterior_marginals(spec, x_tr, s_tr, y_tr, x_tr, 0.3)
    np.testing.assert_allclose(m, md, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd ** 2, vd, rtol=1e-5, atol=1e-7)
    mo, vo = oc.posterior_marginals(spec, x_tr, s_tr, y_tr, None, 0.3)
    np.testing.assert_allclose(m, mo, rtol=1e-8, atol=1e-8)      # vs the oracle's restatement of the same path
    np.testing.assert_allclose(sd ** 2, vo, rtol=1e-8, atol=1e-9)
    # logpdf and rand at the same inputs (:48-78): the mirror's pair-statistic / one-launch routes against the joined 2T-step route and the dense GP
    y_same, s_same = rng.standard_normal(30), rng.random(30) * 0.1 + 0.02
    for s_new in (s_same, 0.15):
        lp = P.logpdf(fpost(x_tr, s_new), y_same)
        lp_d = dg.posterior_logpdf(spec, x_tr, s_tr, y_tr, x_tr, s_new, y_same)
        assert abs(lp - lp_d) <= 1e-6 * abs(lp_d)
        assert abs(lp - fpost(x_tr, s_new)._logpdf_merged(y_same)) <= 1e-8 * abs(lp_d)
    y_gap = y_same.copy()
    y_gap[[0, 7]] = np.nan
    assert abs(P.logpdf(fpost(x_tr, 0.15), y_gap) - fpost(x_tr, 0.15)._logpdf_merged(y_gap)) <= 1e-8 * abs(lp_d)
    ys = P.rand(np.random.default_rng(5), fpost(x_tr, 0.15))
    assert ys.shape == (30,) and np.all(np.isfinite(ys))
    # new inputs (posterior_lti_sde.jl:19-26): merge + sort + missing
    x_pr = np.sort(rng.random(9)) * 6 - 0.5
    m, sd = P.marginals(fpost(x_pr, 0.2))
    md, vd = dg.posterior_marginals(spec, x_tr, s_tr, y_tr, x_pr, 0.2)
    np.testing.assert_allclose(m, md, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd ** 2, vd, rtol=1e-5, atol=1e-7)
    # logpdf of the posterior at new inputs (posterior_lti_sde.jl:62-78)
    y_pr = rng.standard_normal(9)
    s_pr = rng.random(9) * 0.1 + 0.1
    lp = P.logpdf(fpost(x_pr, s_pr), y_pr)
    lp_d = dg.posterior_logpdf(spec, x_tr, s_tr, y_tr, x_pr, s_pr, y_pr)
    assert abs(lp - lp_d) <= 1e-5 * abs(lp_d)
    assert abs(lp - oc.posterior_logpdf(spec, x_tr, s_tr, y_tr, x_pr, s_pr, y_pr)) <= 1e-7 * abs(lp_d)
    # rand of the posterior at new inputs: shape + the same draw through the oracle given the same noise
    ys = P.rand(np.random.default_rng(11), fpost(x_pr, 0.2))
    assert ys.shape == (9,) and np.all(np.isfinite(ys))
    g = np.random.default_rng(11)
    T, d = 39, len(oc.build_lgssm(spec, x_tr, s_tr)["x0m"])
    eps_t, eps_e = g.standard_normal((T, d)), g.standard_normal(T)
    eps_0 = g.standard_normal(d)
    np.testing.assert_allclose(ys, oc.posterior_rand(spec, x_tr, s_tr, y_tr, x_pr, 0.2, eps_t, eps_e, eps_0), rtol=1e-7, atol=1e-7)


def test_rand_many_and_errors(P):
    rng = np.random.default_rng(0)
    fx = P.to_sde(P.GP(P.Matern32Kernel()))(P.RegularSpacing(0.0, 0.1, 25), 0.1)
    Y = P.rand(rng, fx, 4)
    assert Y.shape == (25, 4)
    with pytest.raises(ValueError, match="Dimension mismatch"):
        P.logpdf(fx, np.zeros(24))


@pytest.mark.parametrize("kname", ["base-Matern12", "base-Matern32", "base-Matern52", "scaled-10.0", "stretched-0.1", "sum-12-32",
                                   "sum-32-52-const", "prod-52-32"])
def test_device_side_components_for_irregular_spacing(P, kname):
    """tgp_model_set_sde: A_k = exp(F dt_k), Q_k built on the device from the time stamps (lti_sde.jl:135-146) must give
    the same model as the host construction / the oracle, for every kernel expression (one LTI SDE each)."""
    rng = np.random.default_rng(21)
    spec = KERNELS[kname]
    N = 3000
    x = np.cumsum(rng.random(N) * 0.4 + 1e-3)
    s2 = rng.random(N) * 0.2 + 0.1
    k = P.to_kernel(spec)
    m_dev = P.build_lgssm(k, x, s2, P.ConstMean(0.3), device_components=True)
    m_host = P.build_lgssm(k, x, s2, P.ConstMean(0.3), device_components=False)
    import temporalgps_jl_amd as tgp
    assert isinstance(m_dev.transitions, tgp.lgssm.SDETransitions) and not isinstance(m_host.transitions, tgp.lgssm.SDETransitions)
    y = rng.standard_normal(N)
    ym = y.copy()
    ym[rng.random(N) < 0.1] = np.nan
    lp_o = oc.gp_logpdf(spec, x, s2, y, ("const", 0.3), np.isnan(ym))
    for m in (m_dev, m_host):
        assert abs(tgp.logpdf(m, ym) - lp_o) <= 1e-9 * abs(lp_o)
    a, b = tgp.posterior_marginals(m_dev, y, np.array([0.05])), tgp.posterior_marginals(m_host, y, np.array([0.05]))
    np.testing.assert_allclose(a[0], b[0], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-8, atol=1e-9)
    fa, fb = tgp._filter(m_dev, y), tgp._filter(m_host, y)
    np.testing.assert_allclose(fa[0], fb[0], rtol=1e-8, atol=1e-9)
    pa, pb = tgp.posterior(m_dev, y), tgp.posterior(m_host, y)
    np.testing.assert_allclose(pa.transitions.As, pb.transitions.As, rtol=1e-8, atol=1e-9)
    eps = (rng.standard_normal((N, m_dev.dim)), rng.standard_normal(N), rng.standard_normal(m_dev.dim))
    # Q_k = P - A P A' is a cancellation: at tiny dt two correctly-rounded exponentials give Q's that differ by ~1e-16 |P|,
    # which chol(Q + 1e-9 I) turns into ~1e-9 in a sample
    np.testing.assert_allclose(tgp.rand(eps, m_dev), tgp.rand(eps, m_host), rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("kname,d", [("base-Matern52", 3), ("sum-12-32", 3)])
def test_posterior_consumers_at_the_training_inputs_on_a_regular_grid(P, kname, d, monkeypatch):
    """logpdf / rand of posterior(fx, y)(fx.x, s) (posterior_lti_sde.jl:48-78) on a regular grid: both run on the prior's stationary structure
    (logpdf: two stationary-gain calls through the pair statistic; rand: tgp_posterior_rand) and equal the joined 2T-step route / have its moments."""
    import ctypes
    rng = np.random.default_rng(21)
    T = 20_000
    spec = KERNELS[kname]
    x = P.RegularSpacing(0.0, 0.05, T)
    f = P.to_sde(P.GP(P.to_kernel(spec)), P.HIPStorage())
    fx = f(x, 0.4)
    y = P.rand(rng, fx)
    fpost = P.posterior(fx, y)
    ys = y + 0.3 * rng.standard_normal(T)
    built, real = [], P.build_lgssm
    monkeypatch.setattr(P, "build_lgssm", lambda *a, **k: built.append(real(*a, **k)) or built[-1])

    def stationary_steps(dm):
        hd = dm.handle()
        a, b = ctypes.c_int64(), ctypes.c_int64()
        hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value

    lp = P.logpdf(fpost(x, 0.25), ys)
    assert len(built) == 1 and built[0].T == T and stationary_steps(built[0]) > T - 700     # ONE model of T steps bound, its calls on the stationary engine
    del built[:]
    lp_m = fpost(x, 0.25)._logpdf_merged(ys)
    # (the joined route filters T steps of dt = 0 with singular predicted covariances and carries the reference's own jitter there: it sits 1-2e-9
    #  relative from the dense GP, the pair-statistic route 1e-15 -- measured at T = 2000 / 6000 -- so the two agree to the joined route's error)
    assert abs(lp - lp_m) <= 1e-7 * abs(lp_m)
    xs, k = x.collect()[:1500], slice(0, 1500)
    fp_s = P.posterior(f(xs, 0.4), y[k])(xs, 0.25)
    lp_d = dg.posterior_logpdf(spec, xs, 0.4, y[k], xs, 0.25, ys[k])
    assert abs(P.logpdf(fp_s, ys[k]) - lp_d) <= 1e-12 * abs(lp_d)
    # a draw: its residual against the posterior mean has the posterior's variance (a z-score over T steps), and the emission noise is in it
    m, sd = P.marginals(fpost(x, 0.25))
    del built[:]
    draw = P.rand(np.random.default_rng(1), fpost(x, 0.25))
    assert len(built) == 1 and built[0].T == T and stationary_steps(built[0]) > T - 700
    z = (draw - m) / sd
    assert abs(np.mean(z * z) - 1.0) < 0.05 and abs(np.mean(z)) < 0.1
