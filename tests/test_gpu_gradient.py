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


# ------------------------------------------------------------------------------------------------ adjoint (reverse-time) gradient
ADJ_CASES = {"matern32": [("matern32", 1.3, 0.8)], "matern52": [("matern52", 0.9, 1.2)],
             "sum52_32_12": [("matern52", 1.3, 0.8), ("matern32", 0.6, 1.7), ("matern12", 0.4, 0.5)]}      # 6 kernel parameters + noise, d = 6


def _adj_fx(P, terms, dt, T, noise):
    ks = [P.ScaledKernel(s2, P.StretchedKernel(s, P.to_kernel((name,)))) for name, s2, s in terms]
    k = ks[0]
    for kk in ks[1:]:
        k = k + kk
    names = (["kernel.sigma2", "kernel.kernel.s"] if len(ks) == 1 else
             [f"kernel.kernels[{i}].{leaf}" for i in range(len(ks)) for leaf in ("sigma2", "kernel.s")]) + ["noise"]
    return P.to_sde(P.GP(k))(P.RegularSpacing(0.0, dt, T), noise), names


@pytest.mark.parametrize("case", list(ADJ_CASES))
def test_adjoint_and_tangent_gradients_vs_mpmath_gradient_of_the_oracle(case):
    """d logpdf / d (every kernel variance, every inverse lengthscale, the noise variance) by ONE adjoint pass (tgp_logpdf_adjoint +
    exact block tangents), and by the tangent scans, against the 50-digit gradient of the oracle's recursion (oracle/lgssm_mp.py:
    the whole map hyper-parameters -> logpdf in mpmath, central differences with step 1e-20). Tolerance 1e-8 relative to the
    largest gradient entry."""
    import temporalgps_jl_amd as tgp  # noqa: F401
    from temporalgps_jl_amd import lti_sde as P
    from oracle import lgssm_mp as M
    terms = ADJ_CASES[case]
    T, dt, noise = 640, 0.2, 0.25
    spec = tuple(("scaled", s2, ("stretched", s, (name,))) for name, s2, s in terms)
    model = oc.build_lgssm(spec[0] if len(spec) == 1 else ("sum",) + spec, ("regular", 0.0, dt, T), noise)
    d = len(model["x0m"])
    rng = np.random.default_rng(7)
    y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    lp_mp, g_mp = M.lti_gradient_mp(terms, dt, noise, y)
    scale = max(abs(v) for v in g_mp)
    for method in ("adjoint", "tangent"):
        fx, names = _adj_fx(P, terms, dt, T, noise)
        lp, g = P.logpdf_and_gradient(fx, y, method=method)
        assert abs(lp - lp_mp) <= 1e-10 * abs(lp_mp)
        assert list(g) == names
        for name, want in zip(names, g_mp):
            assert abs(g[name] - want) <= 1e-8 * scale, (method, name, g[name], want)
    fx, _ = _adj_fx(P, terms, dt, T, noise)
    assert P.logpdf_and_gradient(fx, y)[1] == P.logpdf_and_gradient(fx, y, method="adjoint")[1]      # the default policy takes the adjoint pass


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_adjoint_block_gradients_vs_oracle_finite_differences(d):
    """tgp_logpdf_adjoint on random LTI models: the directional derivative along a random direction in the space of ALL blocks
    (A, a, Q, H, h, R, x0m, x0P; symmetric directions for Q and x0P) against central differences of the oracle's sequential logpdf."""
    import temporalgps_jl_amd as tgp
    from tests import _util as U
    rng = np.random.default_rng(100 + d)
    T = 5_003
    model = U.random_lgssm(rng, False, d, T)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    from temporalgps_jl_amd import lgssm as L
    lp, g = L.logpdf_adjoint(dm, y)
    lp_ref = sk.logpdf(model, y)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    sym = lambda X: 0.5 * (X + X.T)
    for trial in range(3):
        dirs = dict(A=rng.standard_normal((d, d)), a=rng.standard_normal(d), Q=sym(rng.standard_normal((d, d))), H=rng.standard_normal(d),
                    h=rng.standard_normal(), R=rng.standard_normal(), x0m=rng.standard_normal(d), x0P=sym(rng.standard_normal((d, d))))
        want = sum(float(np.sum(np.asarray(g[k]) * np.asarray(v))) for k, v in dirs.items())
        gross = sum(float(np.sum(np.abs(np.asarray(g[k]) * np.asarray(v)))) for k, v in dirs.items())      # before the terms cancel
        eps = 1e-6

        def shifted(sgn):
            m = dict(model)
            for k in ("A", "a", "Q", "H"):
                m[k] = model[k] + sgn * eps * dirs[k][None]
            m["h"] = model["h"] + sgn * eps * dirs["h"]
            m["R"] = model["R"] + sgn * eps * dirs["R"]
            m["x0m"] = model["x0m"] + sgn * eps * dirs["x0m"]
            m["x0P"] = model["x0P"] + sgn * eps * dirs["x0P"]
            return sk.logpdf(m, y)
        fd = (shifted(1.0) - shifted(-1.0)) / (2 * eps)
        # (the finite-difference side limits it: the oracle's logpdf carries ~1e-10 of rounding, divided by 2 eps)
        assert abs(want - fd) <= 2e-6 * max(1.0, abs(fd), gross), (trial, want, fd, gross)


def test_adjoint_equals_tangent_scans_on_a_long_series_and_refuses_what_it_does_not_serve():
    import temporalgps_jl_amd as tgp
    from temporalgps_jl_amd import lti_sde as P
    T, dt = 1_000_003, 0.1
    terms = [("matern52", 1.1, 0.9), ("matern12", 0.5, 2.0)]
    fx, names = _adj_fx(P, terms, dt, T, 0.1)
    rng = np.random.default_rng(8)
    y = rng.standard_normal(T) * 0.8
    lp_a, g_a = P.logpdf_and_gradient(fx, y, method="adjoint")
    lp_t, g_t = P.logpdf_and_gradient(fx, y, method="tangent")
    assert abs(lp_a - lp_t) <= 1e-11 * abs(lp_t)
    scale = max(abs(v) for v in g_t.values())
    for n in names:
        assert abs(g_a[n] - g_t[n]) <= 1e-8 * scale, (n, g_a[n], g_t[n])
    # heteroscedastic noise, missing observations, irregular inputs: not the adjoint pass's models
    fxh = P.to_sde(P.GP(P.Matern32Kernel()))(P.RegularSpacing(0.0, 0.1, 2000), np.linspace(0.1, 0.2, 2000))
    with pytest.raises((NotImplementedError, tgp._lib.Unsupported)):
        P.logpdf_and_gradient(fxh, np.zeros(2000), method="adjoint")
    fxm = P.to_sde(P.GP(P.Matern32Kernel()))(P.RegularSpacing(0.0, 0.1, 2000), 0.1)
    ym = np.zeros(2000)
    ym[5] = np.nan
    with pytest.raises((NotImplementedError, tgp._lib.Unsupported)):
        P.logpdf_and_gradient(fxm, ym, method="adjoint")
    lp, g = P.logpdf_and_gradient(fxm, ym)          # the default policy falls back to the tangent scans
    assert np.isfinite(lp) and set(g) == {"noise"}
    # a short series: the one-launch form of the pass (head on the host, d <= 4) serves it as long as it is longer than the head ...
    fxs = P.to_sde(P.GP(P.Matern32Kernel()))(P.RegularSpacing(0.0, 0.1, 300), 0.1)
    ys = np.random.default_rng(3).standard_normal(300)
    lp_a, g_a = P.logpdf_and_gradient(fxs, ys, method="adjoint")
    lp_t, g_t = P.logpdf_and_gradient(fxs, ys, method="tangent")
    assert abs(lp_a - lp_t) <= 1e-11 * abs(lp_t) and abs(g_a["noise"] - g_t["noise"]) <= 1e-8 * abs(g_t["noise"])
    # ... one shorter than the head is refused by the device, served by the tangent scans under the default policy
    fxs = P.to_sde(P.GP(P.Matern32Kernel()))(P.RegularSpacing(0.0, 0.1, 12), 0.1)
    with pytest.raises(tgp._lib.Unsupported):
        P.logpdf_and_gradient(fxs, np.zeros(12), method="adjoint")
    assert set(P.logpdf_and_gradient(fxs, np.zeros(12))[1]) == {"noise"}
