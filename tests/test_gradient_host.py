"""CPU tier: forward-mode gradient of logpdf (the tgp::ad instantiation of the engine, via tests/hostsim) against
central finite differences of the oracle's logpdf over the GP hyper-parameters (sigma^2, inverse lengthscale,
noise variance) -- the correctness test the reference uses for AD (test/gp/lti_sde.jl:203-206: AD vs finite
differences) -- and against the closed-form dense-GP gradient at small N."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import dense_gp as dg
from oracle import lgssm_ref as ref
from tests import _util as U

BASES = [("matern12",), ("matern32",), ("matern52",), ("sum", ("matern52",), ("matern32",))]


def builder(base, T, dt):
    def build(theta):
        s2, inv_l, noise = theta
        return oc.build_lgssm(("scaled", s2, ("stretched", inv_l, base)), ("regular", 0.0, dt, T), noise)
    return build


@pytest.mark.parametrize("base", BASES)
def test_gradient_vs_finite_differences(base):
    T, dt = 120, 0.2
    theta = np.array([1.3, 0.8, 0.25])
    build = builder(base, T, dt)
    model = build(theta)
    rng = np.random.default_rng(1)
    d = len(model["x0m"])
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    lp = ref.logpdf(model, y)
    for k in range(3):
        lml, dl = U.hostsim_grad(model, U.model_tangent(build, theta, k), y)
        assert abs(lml - lp) <= 1e-10 * abs(lp)
        h = 1e-5 * theta[k]
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        fd = (ref.logpdf(build(tp), y) - ref.logpdf(build(tm), y)) / (2 * h)
        assert abs(dl - fd) <= 2e-6 * max(1.0, abs(fd)), (k, dl, fd)


def test_gradient_vs_dense_gp_closed_form():
    # d/dtheta log N(y; 0, K) = 1/2 tr((alpha alpha' - K^-1) dK/dtheta); dK for the noise variance is I
    T, dt = 60, 0.3
    theta = np.array([0.9, 1.1, 0.3])
    base = ("matern32",)
    build = builder(base, T, dt)
    model = build(theta)
    rng = np.random.default_rng(2)
    y = rng.standard_normal(T)
    x = dt * np.arange(T)
    K = dg.kernelmatrix(("scaled", theta[0], ("stretched", theta[1], base)), x) + theta[2] * np.eye(T)
    Ki = np.linalg.inv(K)
    alpha = Ki @ y
    want_noise = 0.5 * np.trace(np.outer(alpha, alpha) - Ki)
    want_s2 = 0.5 * np.trace((np.outer(alpha, alpha) - Ki) @ (dg.kernelmatrix(("stretched", theta[1], base), x)))
    _, d_noise = U.hostsim_grad(model, U.model_tangent(build, theta, 2), y)
    _, d_s2 = U.hostsim_grad(model, U.model_tangent(build, theta, 0), y)
    assert abs(d_noise - want_noise) <= 1e-6 * abs(want_noise)
    assert abs(d_s2 - want_s2) <= 1e-6 * abs(want_s2)


def test_gradient_with_missing():
    T, dt = 90, 0.2
    theta = np.array([1.0, 1.0, 0.2])
    build = builder(("matern52",), T, dt)
    model = build(theta)
    rng = np.random.default_rng(3)
    y = rng.standard_normal(T)
    missing = rng.random(T) < 0.3
    k = 1
    lml, dl = U.hostsim_grad(model, U.model_tangent(build, theta, k), y, missing=missing)
    h = 1e-5
    tp, tm = theta.copy(), theta.copy()
    tp[k] += h
    tm[k] -= h
    fd = (ref.logpdf_missing(build(tp), y, missing) - ref.logpdf_missing(build(tm), y, missing)) / (2 * h)
    assert abs(dl - fd) <= 2e-6 * max(1.0, abs(fd))
