"""Pins the oracle with the identities the reference's own tests assert:

  * state-space == dense GP        /root/reference/test/gp/lti_sde.jl:87-200 (kernel x mean x spacing x noise grid, N=13)
  * posterior at new inputs        /root/reference/test/gp/posterior_lti_sde.jl:52-89 (rtol 1e-5)
  * missing == analytically marginalised     /root/reference/test/models/missings.jl:62-115
  * Scalar == Small with p = 1     /root/reference/test/models/linear_gaussian_conditionals.jl:117-126
"""
import numpy as np
import pytest

from oracle import components as oc
from oracle import dense_gp as dg
from oracle import lgssm_ref as ref

N = 13
KERNELS = {
    "base-Matern12": ("matern12",),
    "base-Matern32": ("matern32",),
    "base-Matern52": ("matern52",),
    **{f"scaled-{s}": ("scaled", s, ("matern32",)) for s in (1e-1, 1.0, 10.0, 100.0)},
    **{f"stretched-{l}": ("stretched", l, ("matern32",)) for l in (1e-2, 0.1, 1.0, 10.0, 100.0)},
    "prod-52-32": ("stretched", 0.01, ("product", ("scaled", 1.5, ("matern52",)), ("matern32",))),
    "prod-32-52-const": ("product", ("scaled", 3.0, ("matern32",)), ("matern52",), ("constant", 1.0)),
    "sum-12-32": ("sum", ("scaled", 1.5, ("stretched", 0.1, ("matern12",))),
                  ("scaled", 0.3, ("stretched", 1.1, ("matern32",)))),
    "sum-32-52-const": ("sum", ("scaled", 2.0, ("matern32",)), ("scaled", 0.5, ("matern52",)),
                        ("scaled", 1.0, ("constant", 1.0))),
}
MEANS = {"zero": None, "const": ("const", 3.0), "custom": ("custom", lambda x: 2 * x)}


def _inputs(spacing):
    if spacing == "regular":
        return ("regular", 0.0, 0.3, N)
    return 0.0 + 0.3 * np.arange(N)


@pytest.mark.parametrize("kname", list(KERNELS))
@pytest.mark.parametrize("mname", list(MEANS))
@pytest.mark.parametrize("spacing", ["regular", "irregular"])
@pytest.mark.parametrize("noise", ["homo", "hetero"])
def test_state_space_equals_dense_gp(kname, mname, spacing, noise):
    rng = np.random.default_rng(123456)
    k, mean, t = KERNELS[kname], MEANS[mname], _inputs(spacing)
    x = oc.times(t)
    s2 = 0.1 if noise == "homo" else rng.random(N) + 1e-1
    model = oc.build_lgssm(k, t, s2, mean)
    d = len(model["x0m"])
    y = ref.rand(model, rng.standard_normal((N, d)), rng.standard_normal(N), rng.standard_normal(d))
    mu, var = ref.marginals(model)
    mu_d, var_d = dg.marginals(k, x, s2, mean)
    np.testing.assert_allclose(mu, mu_d, rtol=1.5e-8, atol=1e-12)
    np.testing.assert_allclose(var, var_d, rtol=1.5e-8)
    lp, lp_d = ref.logpdf(model, y), dg.logpdf(k, x, s2, y, mean)
    assert abs(lp - lp_d) <= 1.5e-8 * abs(lp_d) + 1e-9


@pytest.mark.parametrize("N_ap", [7, 11])
def test_approx_periodic_close_to_periodic(N_ap):
    # test/gp/lti_sde.jl:113-116: the approximation is compared against the true PeriodicKernel.
    rng = np.random.default_rng(1)
    k = ("approx_periodic", N_ap, 1.0)
    t = ("regular", 0.0, 0.3, N)
    model = oc.build_lgssm(k, t, 0.1)
    assert model["A"].shape[1:] == (2 * N_ap, 2 * N_ap)
    d = 2 * N_ap
    y = ref.rand(model, rng.standard_normal((N, d)), rng.standard_normal(N), rng.standard_normal(d))
    lp, lp_d = ref.logpdf(model, y), dg.logpdf(k, oc.times(t), 0.1, y)
    assert abs(lp - lp_d) <= 1e-6 * abs(lp_d)


@pytest.mark.parametrize("kname", ["base-Matern12", "base-Matern32", "base-Matern52", "sum-32-52-const"])
@pytest.mark.parametrize("same", [True, False])
def test_posterior_marginals_equal_dense_gp(kname, same):
    rng = np.random.default_rng(7)
    k = KERNELS[kname]
    x_tr = np.sort(rng.random(20)) * 5
    s_tr = rng.random(20) * 0.2 + 0.05
    y_tr = rng.standard_normal(20)
    x_pr = x_tr if same else np.sort(rng.random(7)) * 6 - 0.5
    s_pr = 0.3
    mu, var = oc.posterior_marginals(k, x_tr, s_tr, y_tr, None if same else x_pr, s_pr)
    mu_d, var_d = dg.posterior_marginals(k, x_tr, s_tr, y_tr, x_pr, s_pr)
    np.testing.assert_allclose(mu, mu_d, rtol=1e-5, atol=1e-7)       # the reference's own bar
    np.testing.assert_allclose(var, var_d, rtol=1e-5, atol=1e-7)


def test_posterior_logpdf_equals_dense_gp():
    rng = np.random.default_rng(8)
    k = ("scaled", 1.3, ("stretched", 0.7, ("matern52",)))
    x_tr, x_pr = np.sort(rng.random(15)) * 4, np.sort(rng.random(6)) * 4 + 0.01
    s_tr, s_pr = 0.2, rng.random(6) * 0.1 + 0.1
    y_tr, y_pr = rng.standard_normal(15), rng.standard_normal(6)
    lp = oc.posterior_logpdf(k, x_tr, s_tr, y_tr, x_pr, s_pr, y_pr)
    lp_d = dg.posterior_logpdf(k, x_tr, s_tr, y_tr, x_pr, s_pr, y_pr)
    assert abs(lp - lp_d) <= 1e-5 * abs(lp_d)


def _random_lgssm(rng, tv, kind, d, p, T):
    """test/models/model_test_utils.jl:163-263 (random_tv_gmm / random_ti_gmm / random_lgssm)."""
    def psd(n, lo, hi):
        U = np.linalg.qr(rng.standard_normal((n, n)))[0]
        return (U * (rng.random(n) * (hi - lo) + lo)) @ U.T
    x0m, x0P = rng.standard_normal(d), psd(d, 0.9, 1.1)
    if tv:
        A = rng.standard_normal((T, d, d))
        a = rng.standard_normal((T, d))
        # the reference's x0.P - A x0.P A' + I is not PSD for a random A (its own comment says so);
        # draw PSD Qs directly instead.
        Q = np.stack([psd(d, 0.5, 1.5) for _ in range(T)])
    else:
        A = -psd(d, 0.1, 0.3)[None]
        a = rng.standard_normal((1, d))
        Q = (x0P - A[0] @ x0P @ A[0].T)[None]
    n = T if tv else 1
    if kind == "scalar":
        H, h, R = rng.standard_normal((n, d)), rng.standard_normal(n), rng.random(n) + 0.1
    else:
        H, h = rng.standard_normal((n, p, d)), rng.standard_normal((n, p))
        R = np.stack([psd(p, 0.9, 1.1) for _ in range(n)])
    return dict(ordering="F", kind=kind, T=T, A=A, a=a, Q=Q, H=H, h=h, R=R, x0m=x0m, x0P=x0P)


@pytest.mark.parametrize("tv", [True, False])
@pytest.mark.parametrize("kind", ["scalar", "small"])
def test_missing_equals_marginalised(tv, kind):
    rng = np.random.default_rng(123456)
    T, d, p = 5, 3, 2
    model = _random_lgssm(rng, tv, kind, d, p, T)
    pp = () if kind == "scalar" else (p,)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal((T,) + pp), rng.standard_normal(d))
    miss_idx, pres_idx = [1, 3], [0, 2, 4]      # Julia [2, 4]
    missing = np.zeros(T, dtype=bool)
    missing[miss_idx] = True
    g = lambda arr, t: arr[t] if arr.shape[0] > 1 else arr[0]
    A, a, Q = [], [], []
    for n in range(T):
        An, an, Qn = g(model["A"], n), g(model["a"], n), g(model["Q"], n)
        if n - 1 in miss_idx:
            Ap, ap, Qp = g(model["A"], n - 1), g(model["a"], n - 1), g(model["Q"], n - 1)
            A.append(An @ Ap); a.append(An @ ap + an); Q.append(An @ Qp @ An.T + Qn)
        else:
            A.append(An); a.append(an); Q.append(Qn)
    new = dict(model)
    sel = lambda arr: np.stack([g(arr, n) for n in pres_idx])
    new.update(T=3, A=np.stack(A)[pres_idx], a=np.stack(a)[pres_idx], Q=np.stack(Q)[pres_idx],
               H=sel(model["H"]), h=sel(model["h"]), R=sel(model["R"]))
    new_y = y[pres_idx]
    assert np.isclose(ref.logpdf(new, new_y), ref.logpdf_missing(model, y, missing), rtol=1e-8)
    fm, fP = ref.filter_(new, new_y)
    gm, gP = ref.filter_missing(model, y, missing)
    np.testing.assert_allclose(fm, gm[pres_idx], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(fP, gP[pres_idx], rtol=1e-7, atol=1e-9)
    pm, pc = ref.marginals(ref.posterior(new, new_y))
    qm, qc = ref.marginals(ref.posterior_missing(model, y, missing))
    np.testing.assert_allclose(pm, qm[pres_idx], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pc, qc[pres_idx], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("tv", [True, False])
@pytest.mark.parametrize("ordering", ["F", "R"])
def test_scalar_equals_small_p1(tv, ordering):
    rng = np.random.default_rng(3)
    T, d = 9, 3
    ms = _random_lgssm(rng, tv, "scalar", d, 1, T)
    ms["ordering"] = ordering
    mv = dict(ms, kind="small", H=ms["H"][:, None, :], h=ms["h"][:, None], R=ms["R"][:, None, None])
    y = rng.standard_normal(T)
    assert np.isclose(ref.logpdf(ms, y), ref.logpdf(mv, y[:, None]), rtol=1e-12)
    a, b = ref.filter_(ms, y), ref.filter_(mv, y[:, None])
    np.testing.assert_allclose(a[0], b[0], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-11, atol=1e-13)
    pa, pb = ref.posterior(ms, y), ref.posterior(mv, y[:, None])
    for key in ("A", "a", "Q", "x0m", "x0P"):
        np.testing.assert_allclose(pa[key], pb[key], rtol=1e-10, atol=1e-12)
    m1, v1 = ref.marginals(pa)
    m2, v2 = ref.marginals(pb)
    np.testing.assert_allclose(m1, m2[:, 0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v1, v2[:, 0, 0], rtol=1e-10, atol=1e-12)


def test_posterior_dimension_mismatch_raises():
    # lgssm.jl:202-208
    model = _random_lgssm(np.random.default_rng(0), False, "scalar", 2, 1, 4)
    with pytest.raises(ValueError, match="Dimension mismatch"):
        ref.posterior(model, np.zeros(5))


@pytest.mark.parametrize("kname", ["base-Matern52", "sum-12-32"])
def test_reverse_ordering_of_a_stationary_lti_model_is_the_forward_model_on_the_flipped_series(kname):
    """gauss_markov_model.jl:38-40 / lgssm.jl:147-165: a Reverse model observes, then predicts; a Forward one predicts, then observes.  When x0
    is stationary under (A, a, Q) -- every to_sde model: x0 = the SDE's stationary distribution -- the Forward model's first predict changes
    nothing, so logpdf(Reverse, y) == logpdf(Forward, flip(y)) and the prior marginals are each other's flips: the route by which Reverse LTI
    priors can reach the stationary-gain engines (not built)."""
    rng = np.random.default_rng(0)
    T = 200
    model = oc.build_lgssm(KERNELS[kname], ("regular", 0.0, 0.3, T), 0.2)
    d = len(model["x0m"])
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    rev = dict(model, ordering="R")
    a, b = ref.logpdf(rev, y), ref.logpdf(model, y[::-1].copy())
    assert abs(a - b) <= 1e-13 * abs(a)
    (mr, vr), (mf, vf) = ref.marginals(rev), ref.marginals(model)
    np.testing.assert_allclose(np.asarray(vr), np.asarray(vf)[::-1], rtol=1e-13)
    np.testing.assert_allclose(np.asarray(mr), np.asarray(mf)[::-1], rtol=1e-13, atol=1e-15)
