"""GPU tier: irregularly spaced inputs with the transitions evaluated INSIDE the passes (TGP_OPT_SDE_CLOSED_FORM; ModelView::sde) --
A_k = exp(F dt_k) in closed form per Matern block and Q_k = P_inf - A_k P_inf A_k' from the 8-byte gap, instead of a tiled [T][2 d^2]
record (reference: broadcast_components for AbstractVector inputs, /root/reference/src/gp/lti_sde.jl:135-146). Checked against the
sequential C oracle on host-built per-step blocks (scipy expm) and against the tiled-record path of the same handle."""
import numpy as np
import pytest
from scipy.linalg import expm

from oracle import components as oc
from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk

pytestmark = pytest.mark.gpu

# sums of scaled / stretched Matern terms, d = 1 .. 8: (name, variance, stretch)
SUMS = {
    1: [("matern12", 1.3, 0.8)],
    2: [("matern32", 0.7, 1.4)],
    3: [("matern52", 1.0, 1.0)],
    "3b": [("matern12", 0.5, 2.0), ("matern32", 1.1, 0.6)],
    4: [("matern52", 1.2, 0.9), ("matern12", 0.4, 1.7)],
    5: [("matern52", 1.0, 1.1), ("matern32", 0.5, 1.5)],
    6: [("matern52", 1.0, 1.0), ("matern52", 0.5, 1.5)],
    7: [("matern52", 0.9, 0.8), ("matern32", 0.6, 1.3), ("matern32", 0.3, 2.1)],
    8: [("matern52", 0.9, 0.8), ("matern52", 0.6, 1.3), ("matern32", 0.3, 2.1)],
}


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd
    return temporalgps_jl_amd


def _kernel(P, terms):
    ks = [P.ScaledKernel(s2, P.StretchedKernel(s, P.to_kernel((nm,)))) for nm, s2, s in terms]
    k = ks[0]
    for kk in ks[1:]:
        k = k + kk
    return k


def _spec(terms):
    spec = tuple(("scaled", s2, ("stretched", s, (nm,))) for nm, s2, s in terms)
    return spec[0] if len(spec) == 1 else ("sum",) + spec


def _times(rng, T, kind):
    if kind == "uniform":
        return np.cumsum(rng.uniform(0.05, 0.15, T))
    if kind == "wild":       # gaps over nine decades: tau^2 N^2 terms both negligible and dominant, exp(-lambda tau) down to underflow
        return np.cumsum(np.exp(rng.uniform(np.log(1e-7), np.log(3e2), T)))
    raise ValueError(kind)


@pytest.mark.parametrize("key", list(SUMS))
@pytest.mark.parametrize("spacing", ["uniform", "wild"])
def test_closed_form_against_oracle_and_tiled_record(tgp, key, spacing):
    from temporalgps_jl_amd import lti_sde as P
    terms = SUMS[key]
    rng = np.random.default_rng(100 * list(SUMS).index(key) + (spacing == "wild"))
    T = 5000
    x = _times(rng, T, spacing)
    s2 = rng.random(T) * 0.2 + 0.05 if key in (2, 5) else 0.1           # per-step noise through the staged stream on two of them
    k = _kernel(P, terms)
    y = rng.standard_normal(T)
    ym = y.copy()
    if key in (3, 6):
        ym[rng.random(T) < 0.1] = np.nan
    Rn = np.array([0.03])
    lp_o = oc.gp_logpdf(_spec(terms), x, s2, y, None, np.isnan(ym))
    out = {}
    for cf in (1, 0):
        m = P.build_lgssm(k, x, s2, device_components=True)
        hd = m.handle()
        hd.set_option(tgp._lib.OPT_SDE_CLOSED_FORM, cf)
        hd.set_option(tgp._lib.OPT_CHUNK, 3)                               # 1667 chunks: two scan levels
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        lp = tgp.logpdf(m, ym)
        names = set(hd.profile())
        assert ("k_tile_dt" in names) == (cf == 1) and ("k_tile_sde" in names) == (cf == 0), names
        assert abs(lp - lp_o) <= 1e-10 * abs(lp_o), (cf, lp, lp_o)
        pm = tgp.posterior_marginals(m, y, Rn)
        mm = tgp.marginals(m)
        fm = tgp._filter(m, y)
        eps = (rng.standard_normal((T, m.dim)), rng.standard_normal(T), rng.standard_normal(m.dim)) if cf == 1 else out[1][4]
        out[cf] = (lp, pm, mm, fm, eps, tgp.rand(eps, m))
    a, b = out[1], out[0]
    assert abs(a[0] - b[0]) <= 1e-11 * abs(b[0])
    for u, v in ((a[1], b[1]), (a[2], b[2]), (a[3][:2], b[3][:2])):
        np.testing.assert_allclose(u[0], v[0], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(u[1], v[1], rtol=1e-8, atol=1e-10)
    if spacing == "uniform":     # (tiny gaps: chol(Q + 1e-9 I) amplifies the rounding of the cancellation in Q, see test_gpu_gp_api)
        np.testing.assert_allclose(a[5], b[5], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("key", [1, 2, 3, "3b", 4, 6])
@pytest.mark.parametrize("spacing", ["uniform", "wild"])
def test_default_engine_selection_on_the_same_cases(tgp, key, spacing):
    """The same cases with nothing forced: d <= 4 goes to the sweep engine (TGP_OPT_SWEEP, one launch per call: k_sweep<sde,...>), which
    must either serve the call or -- gaps over nine decades: no common forgetting length -- hand it to the general engine's closed-form
    passes; larger d stays on those. Same tolerances as above either way."""
    from temporalgps_jl_amd import lti_sde as P
    terms = SUMS[key]
    d = sum({"matern12": 1, "matern32": 2, "matern52": 3}[t[0]] for t in terms)
    rng = np.random.default_rng(100 * list(SUMS).index(key) + (spacing == "wild"))
    T = 5000
    x = _times(rng, T, spacing)
    s2 = rng.random(T) * 0.2 + 0.05 if key in (2, "3b") else 0.1
    y = rng.standard_normal(T)
    ym = y.copy()
    if key in (3, 4):
        ym[rng.random(T) < 0.1] = np.nan
    Rn = np.array([0.03])
    lp_o = oc.gp_logpdf(_spec(terms), x, s2, y, None, np.isnan(ym))
    m = P.build_lgssm(_kernel(P, terms), x, s2, device_components=True)
    hd = m.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    lp = tgp.logpdf(m, ym)
    info, names = hd.sweep_info(), set(hd.profile())
    assert abs(lp - lp_o) <= 1e-10 * abs(lp_o), (lp, lp_o, info)
    if d <= 4 and spacing == "uniform":
        assert info["served"] == 1 and names == {"k_sweep<sde,logpdf>"}, (info, names)
    elif info["served"] == 0:
        assert "k_tile_dt" in names, names
    hd.profile_reset()
    pm = tgp.posterior_marginals(m, ym, Rn)
    info2, names2 = hd.sweep_info(), set(hd.profile())
    if d <= 4 and spacing == "uniform":
        assert info2["served"] == 1 and names2 == {"k_sweep<sde,posterior>"}, (info2, names2)
    ref_m = P.build_lgssm(_kernel(P, terms), x, s2, device_components=True)
    hr = ref_m.handle()
    hr.set_option(tgp._lib.OPT_SWEEP, 0)
    hr.set_option(tgp._lib.OPT_SDE_CLOSED_FORM, 0)
    pr = tgp.posterior_marginals(ref_m, ym, Rn)
    assert hr.sweep_info()["served"] == 0
    np.testing.assert_allclose(pm[0], pr[0], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(pm[1], pr[1], rtol=1e-8, atol=1e-10)


def _host_model(F, Pinf, H, R, times, ordering, first=None):
    """per-step blocks on the host: A_k = expm(F dt_k), Q_k = Pinf - A_k Pinf A_k', dt_1 := 1 (lti_sde.jl:139) unless `first` is given"""
    T, d = len(times), F.shape[0]
    A, Q = np.zeros((T, d, d)), np.zeros((T, d, d))
    for k in range(T):
        if k == 0 and first is not None:
            A[k], Q[k] = first
            continue
        A[k] = expm(F * (1.0 if k == 0 else times[k] - times[k - 1]))
        Q[k] = Pinf - A[k] @ Pinf @ A[k].T
    return dict(ordering=ordering, A=A, a=np.zeros((1, d)), Q=Q, H=H[None], h=np.zeros(1), R=np.atleast_1d(R), x0m=np.zeros(d), x0P=Pinf)


@pytest.mark.parametrize("ordering", ["F", "R"])
@pytest.mark.parametrize("first", [False, True])
def test_both_orderings_and_first_transition(tgp, ordering, first):
    """straight through SDETransitions: Reverse-ordered priors apply transition k + 1 before emission k (gauss_markov_model.jl:40), the
    first transition is the reference's dt_1 := 1 or the pair (A1, Q1) the caller hands over; lambda = 0 (integrated noise) is a block too."""
    rng = np.random.default_rng(7 + (ordering == "R") + 2 * first)
    T = 4000
    lam = 1.7
    F = np.zeros((6, 6))
    F[:3, :3] = np.array([[0, 1, 0], [0, 0, 1], [-lam**3, -3 * lam**2, -3 * lam]])      # Matern-5/2
    F[3:5, 3:5] = np.array([[0, 1], [-0.81, -1.8]])                                       # Matern-3/2, lambda = 0.9
    F[5, 5] = -0.3
    from scipy.linalg import solve_continuous_lyapunov
    Pinf = np.zeros((6, 6))
    for sl, q in ((slice(0, 3), 2.0), (slice(3, 5), 0.7), (slice(5, 6), 1.1)):        # the stationary covariance of each block (Q_k >= 0)
        n = sl.stop - sl.start
        L = np.zeros(n)
        L[-1] = 1.0
        Pinf[sl, sl] = solve_continuous_lyapunov(F[sl, sl], -q * np.outer(L, L))
    H = rng.standard_normal(6)
    times = np.cumsum(rng.uniform(0.02, 0.4, T))
    fst = None
    if first:
        A1 = np.zeros((6, 6))
        for sl, tau in ((slice(0, 3), 0.7), (slice(3, 5), 1.9), (slice(5, 6), 0.4)):    # each term its own dt_1 (stretched kernels)
            A1[sl, sl] = expm(F[sl, sl] * tau)
        fst = (A1, Pinf - A1 @ Pinf @ A1.T)
    model = _host_model(F, Pinf, H, 0.2, times, ordering, fst)
    model.update(kind="scalar", T=T)
    eps = (rng.standard_normal((T, 6)), rng.standard_normal(T), rng.standard_normal(6))
    if ordering == "F":
        y = sk.rand(model, *eps)
        lp_o = sk.logpdf(model, y)
        pm, pv = sk.posterior_marginals(model, y, np.array([0.05]))
    else:                      # (the sequential C oracle runs Forward models; the literal restatement takes both)
        y = np.asarray(ref.rand(model, *eps)).reshape(T)
        lp_o = ref.logpdf(model, y)
        pm, pv = ref.marginals(model)
    order = tgp.Forward if ordering == "F" else tgp.Reverse
    tr = tgp.lgssm.SDETransitions(order, F, times, tgp.Gaussian(np.zeros(6), Pinf), None if fst is None else fst[0], None if fst is None else fst[1])
    for cf in (1, 0):
        m = tgp.LGSSM(tr, tgp.ScalarOutputLGC(H[None], np.zeros(1), np.array([0.2])), T=T)
        hd = m.handle()
        hd.set_option(tgp._lib.OPT_SDE_CLOSED_FORM, cf)
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        lp = tgp.logpdf(m, y)
        assert ("k_tile_dt" in set(hd.profile())) == (cf == 1)
        assert abs(lp - lp_o) <= 1e-10 * abs(lp_o), (cf, lp, lp_o)
        gm, gv = tgp.posterior_marginals(m, y, np.array([0.05])) if ordering == "F" else tgp.marginals(m)
        np.testing.assert_allclose(np.asarray(gm).reshape(-1), np.asarray(pm).reshape(-1), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(np.asarray(gv).reshape(-1), np.asarray(pv).reshape(-1), rtol=1e-8, atol=1e-9)


def test_other_drift_matrices_keep_the_tiled_record(tgp):
    """complex eigenvalues (a damped oscillator), distinct real eigenvalues inside one block, blocks longer than 3: no closed form of
    this kind -- the tiled record serves them, same results as the host-built blocks."""
    rng = np.random.default_rng(3)
    T = 3000
    times = np.cumsum(rng.uniform(0.05, 0.3, T))
    cases = {
        "oscillator": np.array([[0.0, 1.0], [-4.0, -0.6]]),
        "two real": np.array([[-1.0, 0.3], [0.2, -2.0]]),
        "chain of four": np.diag(np.ones(3), 1) + np.vstack([np.zeros((3, 4)), -np.array([[1.0, 4.0, 6.0, 4.0]])]),   # (s + 1)^4
    }
    for name, F in cases.items():
        d = F.shape[0]
        B = rng.standard_normal((d, d))
        Pinf = B @ B.T + np.eye(d)
        if name == "chain of four":      # exp(F dt) contracts P_inf only if P_inf solves the Lyapunov equation: take that one
            from scipy.linalg import solve_continuous_lyapunov
            L = np.zeros(d)
            L[-1] = 1.0
            Pinf = solve_continuous_lyapunov(F, -np.outer(L, L))
        else:
            from scipy.linalg import solve_continuous_lyapunov
            Pinf = solve_continuous_lyapunov(F, -np.eye(d))
        H = rng.standard_normal(d)
        model = _host_model(F, Pinf, H, 0.3, times, "F")
        model.update(kind="scalar", T=T)
        y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        lp_o = sk.logpdf(model, y)
        tr = tgp.lgssm.SDETransitions(tgp.Forward, F, times, tgp.Gaussian(np.zeros(d), Pinf))
        m = tgp.LGSSM(tr, tgp.ScalarOutputLGC(H[None], np.zeros(1), np.array([0.3])), T=T)
        hd = m.handle()
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        lp = tgp.logpdf(m, y)
        assert "k_tile_sde" in set(hd.profile()) and "k_tile_dt" not in set(hd.profile()), name
        assert abs(lp - lp_o) <= 1e-10 * abs(lp_o), (name, lp, lp_o)


def test_gradient_call_between_value_calls(tgp):
    """the dual-number passes read A_k, Q_k beside their tangents: the record is rebuilt in the tiled form for them and back for the
    next value call (d <= 4)."""
    from temporalgps_jl_amd import lti_sde as P
    rng = np.random.default_rng(5)
    T = 3000
    x = np.cumsum(rng.uniform(0.05, 0.3, T))
    y = rng.standard_normal(T)
    fx = P.to_sde(P.GP(1.3 * P.Matern52Kernel().stretch(0.9)))(x, 0.25)
    fx_dev = P.to_sde(P.GP(1.3 * P.Matern52Kernel().stretch(0.9)))(x, 0.25)
    P_min = P.DEVICE_COMPONENTS_MIN_T
    try:
        P.DEVICE_COMPONENTS_MIN_T = 1
        lp0 = P.logpdf(fx_dev, y)
        lp1, g = P.logpdf_and_gradient(fx_dev, y)
        lp2 = P.logpdf(fx_dev, y)
    finally:
        P.DEVICE_COMPONENTS_MIN_T = P_min
    assert abs(lp0 - lp1) <= 1e-11 * abs(lp0) and lp0 == lp2
    _, g_fd = P.logpdf_and_gradient(fx, y, method="fd")
    for n in g:
        assert abs(g[n] - g_fd[n]) <= 1e-5 * max(1.0, abs(g_fd[n])), (n, g[n], g_fd[n])


@pytest.mark.parametrize("key", [5, 6, 7, 8])
def test_per_step_noise_runs_the_kernels_of_the_per_step_verdict(tgp, key):
    """d = 5..8 choose between two builds of the passes by a run-time known-answer check with one verdict per layout family. A model
    bound through tgp_model_set_sde belongs to the per-step family whatever its emission flags say: until round 3 it got the LTI
    family's verdict when the noise was per-step, and at d = 8 with more than 256 chunks ran inlined kernels the per-step check had
    rejected (log-likelihood wrong in the fourth digit, or a spurious not-positive-definite error; scripts/stress_sde.py, seed 2).
    No option is touched here: options re-select the table."""
    from temporalgps_jl_amd import lti_sde as P
    terms = SUMS[key]
    rng = np.random.default_rng(40 + key)
    T = 4097
    x = np.cumsum(np.exp(rng.normal(np.log(0.1), 2.0 / 3.0, T)))
    s2 = rng.random(T) * 0.3 + 0.02
    y = rng.standard_normal(T)
    lp_o = oc.gp_logpdf(_spec(terms), x, s2, y, None, None)
    m = P.build_lgssm(_kernel(P, terms), x, s2, device_components=True)
    lp = tgp.logpdf(m, y)
    assert abs(lp - lp_o) <= 1e-10 * abs(lp_o), (lp, lp_o)
    m2 = P.build_lgssm(_kernel(P, terms), x, s2, device_components=False)       # host-built per-step blocks: the same family
    lp2 = tgp.logpdf(m2, y)
    assert abs(lp2 - lp_o) <= 1e-10 * abs(lp_o), (lp2, lp_o)


def test_repeated_and_unsorted_time_stamps(tgp):
    """two observations at the same time (dt = 0: A = I, Q = 0 in both forms) are a valid model; time stamps out of order are refused at
    tgp_model_set_sde (the reference would exponentiate a negative gap)."""
    from temporalgps_jl_amd import lti_sde as P
    rng = np.random.default_rng(2)
    T = 2000
    x = np.cumsum(rng.uniform(0.05, 0.15, T))
    x[100] = x[99]
    x[1500] = x[1499]
    y = rng.standard_normal(T)
    spec = ("scaled", 1.2, ("stretched", 0.8, ("matern52",)))
    lp_o = oc.gp_logpdf(spec, x, 0.1, y, None, None)
    for cf in (1, 0):
        m = P.build_lgssm(P.to_kernel(spec), x, 0.1, device_components=True)
        m.handle().set_option(tgp._lib.OPT_SDE_CLOSED_FORM, cf)
        assert abs(tgp.logpdf(m, y) - lp_o) <= 1e-10 * abs(lp_o)
    xb = x.copy()
    xb[700], xb[701] = xb[701] + 0.01, xb[700]
    with pytest.raises(tgp._lib.TGPError, match="non-decreasing"):
        P.build_lgssm(P.to_kernel(spec), xb, 0.1, device_components=True).handle()
