"""GPU tier: d logpdf / d theta by forward-mode tangent scans (tgp_logpdf_grad) against central finite differences of
the ORACLE's logpdf (the reference's own AD correctness test is AD vs finite differences, test/gp/lti_sde.jl:203-206)
and against the closed-form dense-GP gradient. Tolerance 2e-6 relative (the finite-difference side limits it)."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import dense_gp as dg
from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk

pytestmark = pytest.mark.gpu
BASES = {"matern12": ("matern12",), "matern32": ("matern32",), "matern52": ("matern52",),
         "sum52_32": ("sum", ("matern52",), ("matern32",)), "sum52_52": ("sum", ("matern52",), ("matern52",))}


def product_kernel(P, base, s2, inv_l):
    return P.ScaledKernel(s2, P.StretchedKernel(inv_l, P.to_kernel(base)))


@pytest.mark.parametrize("bname", list(BASES))
@pytest.mark.parametrize("T,chunk", [(300, 2), (20000, 0)])
def test_gradient_vs_oracle_finite_differences(bname, T, chunk):
    import temporalgps_jl_amd as tgp
    from temporalgps_jl_amd import lti_sde as P
    base = BASES[bname]
    theta = np.array([1.3, 0.8, 0.25])
    dt = 0.2
    build = lambda th: oc.build_lgssm(("scaled", th[0], ("stretched", th[1], base)), ("regular", 0.0, dt, T), th[2])
    model = build(theta)
    d = len(model["x0m"])
    rng = np.random.default_rng(1)
    y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    fx = P.to_sde(P.GP(product_kernel(P, base, theta[0], theta[1])))(P.RegularSpacing(0.0, dt, T), theta[2])
    if chunk:
        fx.build_lgssm().handle().set_option(tgp._lib.OPT_CHUNK, chunk)
    lp, g = P.logpdf_and_gradient(fx, y)
    lp_ref = sk.logpdf(model, y)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    got = np.array([g["kernel.sigma2"], g["kernel.kernel.s"], g["noise"]])
    for k in range(3):
        h = 1e-5 * theta[k]
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        fd = (sk.logpdf(build(tp), y) - sk.logpdf(build(tm), y)) / (2 * h)
        assert abs(got[k] - fd) <= 2e-6 * max(1.0, abs(fd)), (k, got[k], fd)


def test_gradient_vs_dense_gp_closed_form_and_missing():
    import temporalgps_jl_amd as tgp  # noqa: F401
    from temporalgps_jl_amd import lti_sde as P
    T, dt = 80, 0.3
    theta = np.array([0.9, 1.1, 0.3])
    base = ("matern32",)
    rng = np.random.default_rng(2)
    y = rng.standard_normal(T)
    x = dt * np.arange(T)
    K = dg.kernelmatrix(("scaled", theta[0], ("stretched", theta[1], base)), x) + theta[2] * np.eye(T)
    Ki = np.linalg.inv(K)
    alpha = Ki @ y
    W = np.outer(alpha, alpha) - Ki
    fx = P.to_sde(P.GP(product_kernel(P, base, theta[0], theta[1])))(P.RegularSpacing(0.0, dt, T), theta[2])
    lp, g = P.logpdf_and_gradient(fx, y)
    assert abs(g["noise"] - 0.5 * np.trace(W)) <= 1e-6 * abs(0.5 * np.trace(W))
    want_s2 = 0.5 * np.trace(W @ dg.kernelmatrix(("stretched", theta[1], base), x))
    assert abs(g["kernel.sigma2"] - want_s2) <= 1e-6 * abs(want_s2)
    # missing observations + a constant mean
    fx = P.to_sde(P.GP(P.ConstMean(0.7), product_kernel(P, base, theta[0], theta[1])))(P.RegularSpacing(0.0, dt, T), theta[2])
    missing = rng.random(T) < 0.3
    ym = y.copy()
    ym[missing] = np.nan
    lp, g = P.logpdf_and_gradient(fx, ym)
    spec = lambda th, c: (oc.build_lgssm(("scaled", th[0], ("stretched", th[1], base)), ("regular", 0.0, dt, T), th[2], ("const", c)))
    f = lambda th, c: ref.logpdf_missing(spec(th, c), y, missing)
    assert abs(lp - f(theta, 0.7)) <= 1e-10 * abs(lp)
    h = 1e-6
    assert abs(g["mean.c"] - (f(theta, 0.7 + h) - f(theta, 0.7 - h)) / (2 * h)) <= 2e-6 * max(1.0, abs(g["mean.c"]))
    tp, tm = theta.copy(), theta.copy()
    tp[1] += h
    tm[1] -= h
    fd = (f(tp, 0.7) - f(tm, 0.7)) / (2 * h)
    assert abs(g["kernel.kernel.s"] - fd) <= 2e-6 * max(1.0, abs(fd))


def test_gradient_unsupported_layouts_raise():
    import temporalgps_jl_amd as tgp  # noqa: F401
    from temporalgps_jl_amd import lti_sde as P
    fx = P.to_sde(P.GP(P.Matern32Kernel()))(np.cumsum(np.ones(10) * 0.1), 0.1)      # irregular spacing => per-step blocks
    with pytest.raises(NotImplementedError):
        P.logpdf_and_gradient(fx, np.zeros(10))
