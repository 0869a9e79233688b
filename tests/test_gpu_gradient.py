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
    # plain-array inputs (the reference's AbstractVector path) are served through the SDE-described model up to d = 4 only
    fx = P.to_sde(P.GP(P.Matern52Kernel() + P.Matern52Kernel().stretch(0.5)))(np.cumsum(np.ones(10) * 0.1), 0.1)     # d = 6
    with pytest.raises(NotImplementedError):
        P.logpdf_and_gradient(fx, np.zeros(10))


# ------------------------------------------------------------------------------------------------ irregular spacing
def _irregular_case(seed, T):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0.0, 0.05 * T, T))
    t += np.arange(T) * 1e-6                       # strictly increasing
    return rng, t


@pytest.mark.parametrize("case", ["matern32", "matern52_mean", "sum52_12", "hetero"])
def test_gradient_irregular_spacing_vs_oracle_fd(case):
    """tgp_logpdf_grad_sde: per-step tangents dA_k, dQ_k built on the device from (dF, dPinf); against central differences of
    the oracle's logpdf on the same irregular inputs (the reference differentiates this path with Mooncake,
    test/gp/lti_sde.jl:203-206)."""
    import temporalgps_jl_amd as tgp  # noqa: F401
    from temporalgps_jl_amd import lti_sde as S
    T = 3000
    rng, t = _irregular_case(11, T)
    y = rng.standard_normal(T)
    th0 = {"matern32": [0.8, 1.7, 0.3], "matern52_mean": [1.3, 0.9, 0.25, 0.4], "sum52_12": [0.7, 1.4, 0.5, 2.0, 0.2],
           "hetero": [1.1, 0.6]}[case]
    noise_vec = 0.2 + 0.3 * rng.random(T)

    def spec(th):
        if case == "matern32":
            return ("scaled", th[0], ("stretched", th[1], ("matern32",))), th[2], None
        if case == "matern52_mean":
            return ("scaled", th[0], ("stretched", th[1], ("matern52",))), th[2], ("const", th[3])
        if case == "sum52_12":
            return ("sum", ("scaled", th[0], ("stretched", th[1], ("matern52",))), ("scaled", th[2], ("stretched", th[3], ("matern12",)))), th[4], None
        return ("scaled", th[0], ("stretched", th[1], ("matern52",))), noise_vec, None

    def product(th):
        k, s2, mean = spec(th)
        gp = S.GP(S.to_kernel(k)) if mean is None else S.GP(float(mean[1]), S.to_kernel(k))
        return S.to_sde(gp)(t, s2)

    def oracle_lp(th):
        k, s2, mean = spec(th)
        return oc.gp_logpdf(k, t, s2, y, mean=mean)

    lp, g = S.logpdf_and_gradient(product(th0), y)
    lp_ref = oracle_lp(th0)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    order = {"matern32": ["kernel.sigma2", "kernel.kernel.s", "noise"],
             "matern52_mean": ["kernel.sigma2", "kernel.kernel.s", "noise", "mean.c"],
             "sum52_12": ["kernel.kernels[0].sigma2", "kernel.kernels[0].kernel.s", "kernel.kernels[1].sigma2", "kernel.kernels[1].kernel.s", "noise"],
             "hetero": ["kernel.sigma2", "kernel.kernel.s"]}[case]
    assert set(g) == set(order), (sorted(g), order)
    for i, name in enumerate(order):
        hstep = 1e-5 * max(1.0, abs(th0[i]))
        tp, tm = list(th0), list(th0)
        tp[i] += hstep
        tm[i] -= hstep
        fd = (oracle_lp(tp) - oracle_lp(tm)) / (2 * hstep)
        assert abs(g[name] - fd) <= 2e-5 * max(1.0, abs(fd)), (name, g[name], fd)


def test_uniform_plain_array_inputs_take_the_vector_path_and_cached_models_follow_parameter_changes():
    """(i) A uniformly spaced plain array is still the reference's AbstractVector input (lti_sde.jl:135-146, dt_1 := 1): its
    gradient goes through the SDE-described model, not the LTI blocks. (ii) The finite GP caches its device model; changing a
    hyper-parameter in place through the handles `parameters` returns re-binds it."""
    from temporalgps_jl_amd import lti_sde as S
    T = 800
    rng = np.random.default_rng(5)
    t = 0.07 * np.arange(T)                       # uniform, but a plain array
    y = rng.standard_normal(T)
    k = ("scaled", 0.9, ("stretched", 1.3, ("matern52",)))
    fx = S.to_sde(S.GP(S.to_kernel(k)))(t, 0.2)
    lp, g = S.logpdf_and_gradient(fx, y)
    lp_ref = oc.gp_logpdf(k, t, 0.2, y)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    eps = 1e-5
    fd = (oc.gp_logpdf(("scaled", 0.9 + eps, k[2]), t, 0.2, y) - oc.gp_logpdf(("scaled", 0.9 - eps, k[2]), t, 0.2, y)) / (2 * eps)
    assert abs(g["kernel.sigma2"] - fd) <= 1e-5 * max(1.0, abs(fd))
    # in-place parameter change
    name, owner, attr = S.parameters(fx.f.f.kernel)[0]
    assert abs(S.logpdf(fx, y) - lp_ref) <= 1e-10 * abs(lp_ref)
    setattr(owner, attr, 1.7)
    lp2_ref = oc.gp_logpdf(("scaled", 1.7, k[2]), t, 0.2, y)
    assert abs(S.logpdf(fx, y) - lp2_ref) <= 1e-10 * abs(lp2_ref)


def test_gradient_policy_from_d9_is_central_differences_of_the_device_logpdf():
    """d >= 9: the dual-number kernels are out-of-line private-memory code (d = 14, T = 2e5: 2.4 s against 5.7 ms per logpdf);
    logpdf_and_gradient then differences the group-kernel logpdf. Same gradient as the tangent scans to ~1e-6."""
    import time
    from temporalgps_jl_amd import lti_sde as S
    T = 3000
    rng = np.random.default_rng(9)
    y = rng.standard_normal(T)
    k = S.Matern52Kernel() + S.Matern52Kernel().stretch(0.5) + 0.5 * S.Matern52Kernel().stretch(2.0)      # d = 9
    fx = S.to_sde(S.GP(0.3, k))(S.RegularSpacing(0.0, 0.05, T), 0.2)
    assert fx.build_lgssm().dim == 9
    t0 = time.perf_counter()
    lp_fd, g_fd = S.logpdf_and_gradient(fx, y)
    t_fd = time.perf_counter() - t0
    lp_t, g_t = S.logpdf_and_gradient(fx, y, method="tangent")
    assert lp_fd == lp_t or abs(lp_fd - lp_t) <= 1e-12 * abs(lp_t)
    assert set(g_fd) == set(g_t)
    for name in g_t:
        assert abs(g_fd[name] - g_t[name]) <= 2e-6 * max(1.0, abs(g_t[name])), (name, g_fd[name], g_t[name])
    assert abs(S.logpdf(fx, y) - lp_t) <= 1e-12 * abs(lp_t)          # the parameters are restored
