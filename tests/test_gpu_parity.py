"""GPU tier (-m gpu): the HIP path, called through the C ABI (ctypes -> libtgp_hip.so), against the oracle
on the same seeded inputs. Tolerances (fp64, north_star "stated fp64 tolerance"):
   logpdf            rel 1e-10 vs the sequential restatement
   filter / posterior states, marginals   abs/rel 1e-8 (the reference's 1e-10 jitter in invert_dynamics
                                          is visible at ~1e-10)
   rand              rel 1e-9 given identical noise
"""
import os

import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk
from tests import _util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def to_device_model(tgp, model):
    assert model["kind"] == "scalar"
    tr = tgp.GaussMarkovModel(tgp.Forward if model["ordering"] == "F" else tgp.Reverse, model["A"], model["a"], model["Q"],
                              tgp.Gaussian(model["x0m"], model["x0P"]))
    return tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])


def check_all(tgp, model, y, eps, chunk=None, missing=None):
    dm = to_device_model(tgp, model)
    if chunk is not None:
        dm.handle().set_option(tgp._lib.OPT_CHUNK, chunk)
    T = model["T"]
    yin = y.copy()
    if missing is not None:
        yin[missing] = np.nan
        lp = ref.logpdf_missing(model, y, missing)
        fm, fP = ref.filter_missing(model, y, missing)
        post = ref.posterior_missing(model, y, missing)
    else:
        lp = ref.logpdf(model, y)
        fm, fP = ref.filter_(model, y)
        post = ref.posterior(model, y) if model["ordering"] == "F" else None
    got = tgp.logpdf(dm, yin)
    assert abs(got - lp) <= 1e-10 * abs(lp), (got, lp)
    m, P = tgp._filter(dm, yin)
    np.testing.assert_allclose(m, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, fP, rtol=1e-8, atol=1e-9)
    if post is not None:
        dpost = tgp.posterior(dm, yin)
        assert dpost.ordering is tgp.Reverse
        np.testing.assert_allclose(dpost.transitions.As, post["A"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.transitions.as_, post["a"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.transitions.Qs, post["Q"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.x0.m, post["x0m"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.x0.P, post["x0P"], rtol=1e-8, atol=1e-9)
        Rn = np.random.default_rng(5).random(T) * 0.1
        pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
        gm, gv = tgp.posterior_marginals(dm, yin, Rn)
        np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv, pv, rtol=1e-8, atol=1e-9)
        # marginals of the MATERIALISED posterior model (Reverse ordering, per-step G, g, L)
        gm2, gv2 = tgp.marginals(tgp.replace_observation_noise_cov(dpost, Rn))
        np.testing.assert_allclose(gm2, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv2, pv, rtol=1e-8, atol=1e-9)
    mm, mv = ref.marginals(model)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(gm, mm, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(gv, mv, rtol=1e-10, atol=1e-11)
    if eps is not None:
        ys = tgp.rand(eps, dm)
        np.testing.assert_allclose(ys, ref.rand(model, *eps), rtol=1e-9, atol=1e-9)


GP_CASES = [
    (("matern12",), ("regular", 0.0, 0.1, 997), 0.1),
    (("matern32",), ("regular", 0.0, 0.1, 1000), 0.1),               # cfg1-like (T reduced for the Python oracle)
    (("matern52",), ("regular", 0.0, 0.1, 1333), 0.1),
    (("sum", ("matern52",), ("matern32",)), ("regular", 0.0, 0.1, 500), 0.1),
    (("sum", ("matern52",), ("matern52",)), ("regular", 0.0, 0.05, 400), 0.2),
    (("sum", ("matern52",), ("matern12",)), ("regular", 0.0, 0.05, 400), 0.2),
    (("scaled", 1.0, ("stretched", 1 / 2.3, ("matern52",))), ("regular", -5.0, 1e-2, 2000), 0.5),   # bench/single_output_gps.jl
]


@pytest.mark.parametrize("i", range(len(GP_CASES)))
@pytest.mark.parametrize("chunk", [None, 1, 3])
def test_gp_regular_spacing(tgp, i, chunk):
    k, t, s2 = GP_CASES[i]
    model, y, eps = U.gp_case(k, t, s2, seed=i)
    check_all(tgp, model, y, eps, chunk=chunk)


def test_gp_irregular_heteroscedastic(tgp):
    rng = np.random.default_rng(42)
    t = np.cumsum(rng.random(800) * 0.1 + 0.05)
    model, y, eps = U.gp_case(("scaled", 1.5, ("stretched", 0.7, ("matern52",))), t, rng.random(800) * 0.2 + 0.05, seed=9,
                              mean=("const", 3.0))
    check_all(tgp, model, y, eps)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("tv", [True, False])
def test_random_lgssm(tgp, d, tv):
    rng = np.random.default_rng(10 * d + tv)
    T = 300
    model = U.random_lgssm(rng, tv, d, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    check_all(tgp, model, ref.rand(model, *eps), eps, chunk=2)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_multilevel_scan_every_d(tgp, d):
    """n0 = 3000 chunk elements -> two scan levels + top, for every compiled state dimension."""
    rng = np.random.default_rng(70 + d)
    T = 6000
    model = U.random_lgssm(rng, False, d, T)
    model["R"] = rng.random(T) + 0.1                      # per-step R through the staged IO
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = sk.rand(model, *eps)
    dm = to_device_model(tgp, model)
    dm.handle().set_option(tgp._lib.OPT_CHUNK, 2)
    lp = sk.logpdf(model, y)
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-10 * abs(lp)
    Rn = rng.random(T) * 0.1
    pm, pv = sk.posterior_marginals(model, y, Rn)
    gm, gv = tgp.posterior_marginals(dm, y, Rn)
    np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(gv, pv, rtol=1e-8, atol=1e-9)
    mm, mv = sk.prior_marginals(model)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(gm, mm, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(gv, mv, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(tgp.rand(eps, dm), y, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tv", [True, False])
def test_missing(tgp, tv):
    rng = np.random.default_rng(3)
    T, d = 700, 3
    model = U.random_lgssm(rng, tv, d, T)
    y = rng.standard_normal(T)
    check_all(tgp, model, y, None, missing=rng.random(T) < 0.3)


@pytest.mark.parametrize("tv", [True, False])
def test_reverse_ordering(tgp, tv):
    rng = np.random.default_rng(11)
    T, d = 400, 3
    model = U.random_lgssm(rng, tv, d, T, ordering="R")
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    check_all(tgp, model, ref.rand(model, *eps), eps)


def test_errors(tgp):
    rng = np.random.default_rng(0)
    model = U.random_lgssm(rng, False, 2, 10)
    dm = to_device_model(tgp, model)
    with pytest.raises(ValueError, match="Dimension mismatch"):       # lgssm.jl:202-208
        tgp.logpdf(dm, np.zeros(11))
    with pytest.raises(ValueError, match="Dimension mismatch"):
        tgp.posterior(dm, np.zeros(9))
    bad = dict(model, R=np.array([-10.0]))                            # S <= 0: Julia throws DomainError (sqrt)
    with pytest.raises(tgp._lib.NotPositiveDefinite):
        tgp.logpdf(to_device_model(tgp, bad), rng.standard_normal(10))


@pytest.mark.parametrize("kname,T", [("matern32", 1_000_000), ("matern52", 1_000_000)])
def test_large_T_vs_c_oracle(tgp, kname, T):
    """Multi-level scans at scale, against the C restatement (same inputs)."""
    model = oc.build_lgssm((kname,), ("regular", 0.0, 0.1, T), 0.1)
    d = len(model["x0m"])
    rng = np.random.default_rng(2)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = sk.rand(model, *eps)
    dm = to_device_model(tgp, model)
    lp_ref = sk.logpdf(model, y)
    for chunk in (None, 2):
        dm.handle().set_option(tgp._lib.OPT_CHUNK, chunk or 0)
        lp = tgp.logpdf(dm, y)
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (chunk, lp, lp_ref)
        assert tgp.logpdf(dm, y) == lp       # fixed-order reductions: bit-reproducible
    mean_ref, var_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
    mean, var = tgp.posterior_marginals(dm, y, np.array([1e-18]))
    assert np.max(np.abs(mean - mean_ref)) <= 1e-8
    assert np.max(np.abs(var - var_ref)) <= 1e-8
    np.testing.assert_allclose(tgp.rand(eps, dm), y, rtol=1e-9, atol=1e-9)


def test_device_resident_inputs(tgp):
    """torch CUDA tensors are used in place (no host copies) and results stay on the device."""
    import torch
    T = 50_000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    rng = np.random.default_rng(4)
    y = sk.rand(model, rng.standard_normal((T, 3)), rng.standard_normal(T), rng.standard_normal(3))
    dm = to_device_model(tgp, model)
    yd = torch.as_tensor(y, device="cuda:0")
    lp = tgp.logpdf(dm, yd)
    assert abs(lp - sk.logpdf(model, y)) <= 1e-10 * abs(lp)
    mean, var = tgp.posterior_marginals(dm, yd, np.array([0.0]))
    assert mean.is_cuda and var.is_cuda
    mr, vr = sk.posterior_marginals(model, y, np.array([0.0]))
    assert np.max(np.abs(mean.cpu().numpy() - mr)) <= 1e-8 and np.max(np.abs(var.cpu().numpy() - vr)) <= 1e-8


@pytest.mark.parametrize("d", [5, 6])
@pytest.mark.parametrize("variant", [1, 2])
def test_kernel_variants_d5_d6(tgp, d, variant):
    """d = 5, 6 exist as an out-of-line (safe) and a fully inlined (fast) build; both must match the oracle, and the
    automatic choice must have passed the library's own run-time known-answer check."""
    rng = np.random.default_rng(500 + d)
    T = 2500
    for tv in (True, False):
        model = U.random_lgssm(rng, tv, d, T)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        y = sk.rand(model, *eps)
        dm = to_device_model(tgp, model)
        hd = dm.handle()
        auto = hd.lib.tgp_kernel_variant(hd.h)
        assert auto in (1, 2)
        hd.set_option(tgp._lib.OPT_VARIANT, variant)
        assert hd.lib.tgp_kernel_variant(hd.h) == variant
        hd.set_option(tgp._lib.OPT_CHUNK, 3)
        lp = sk.logpdf(model, y)
        assert abs(tgp.logpdf(dm, y) - lp) <= 1e-10 * abs(lp)
        Rn = rng.random(T) * 0.1
        pm, pv = sk.posterior_marginals(model, y, Rn)
        gm, gv = tgp.posterior_marginals(dm, y, Rn)
        np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv, pv, rtol=1e-8, atol=1e-9)
        post_c = sk.posterior(model, y)
        post = tgp.posterior(dm, y)
        np.testing.assert_allclose(post.transitions.As, post_c["A"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(post.transitions.Qs, post_c["Q"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(tgp.rand(eps, dm), y, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("d", [9, 13, 16] if os.environ.get("TGP_TEST_ALL_D") == "1" else [9, 16])
@pytest.mark.parametrize("tv", [True, False])
def test_larger_state_dimensions(tgp, d, tv):
    """d = 9..16 (e.g. ApproxPeriodicKernel{7}: d = 14) run the out-of-line, private-memory build."""
    rng = np.random.default_rng(900 + d + tv)
    T = 600
    model = U.random_lgssm(rng, tv, d, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    check_all(tgp, model, ref.rand(model, *eps), eps, chunk=2)


@pytest.mark.parametrize("d", [24, 32])
@pytest.mark.parametrize("tv", [True, False])
def test_larger_state_dimensions_dense_engine(tgp, d, tv):
    """d = 17..32 (and beyond) bind the dense fp64-MFMA engine (tgp_dense.hip): every operation of the interface -- logpdf,
    filter, posterior (G, g, L), posterior marginals (segmented smoother: TGP_OPT_CHUNK = 7), prior marginals, rand -- against
    the oracle, like the scan path's state dimensions above."""
    rng = np.random.default_rng(1900 + d + tv)
    T = 300
    model = U.random_lgssm(rng, tv, d, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    check_all(tgp, model, ref.rand(model, *eps), eps, chunk=7)
    hd = to_device_model(tgp, model).handle()
    assert hd.lib.tgp_kernel_variant(hd.h) >= 16


def test_approx_periodic_default_kernel(tgp):
    """ApproxPeriodicKernel() (7 cosine terms, d = 14; lti_sde.jl:255-307) against the dense GP with the true periodic
    kernel (test/gp/lti_sde.jl:113-116) and against the oracle's state-space restatement."""
    from oracle import dense_gp as dg
    from temporalgps_jl_amd import lti_sde as P
    rng = np.random.default_rng(14)
    N = 300
    spec = ("approx_periodic", 7, 1.0)
    x = P.RegularSpacing(0.0, 0.13, N)
    fx = P.to_sde(P.GP(P.ApproxPeriodicKernel()))(x, 0.1)
    y = P.rand(rng, fx)
    lp = P.logpdf(fx, y)
    lp_o = oc.gp_logpdf(spec, ("regular", 0.0, 0.13, N), 0.1, y)
    assert abs(lp - lp_o) <= 1e-9 * abs(lp_o)
    lp_d = dg.logpdf(spec, x.collect(), 0.1, y)
    assert abs(lp - lp_d) <= 1e-5 * abs(lp_d)                 # 7 terms approximate the periodic kernel to ~1e-7
    m, sd = P.marginals(P.posterior(fx, y)(x, 0.05))
    mo, vo = oc.posterior_marginals(spec, ("regular", 0.0, 0.13, N), 0.1, y, None, 0.05)
    np.testing.assert_allclose(m, mo, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(sd ** 2, vo, rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("T", [1, 2, 7, 8, 9])
@pytest.mark.parametrize("d", [1, 3, 6, 8, 11] if os.environ.get("TGP_TEST_ALL_D") == "1" else [1, 3, 6, 8, 16])
def test_tiny_series(tgp, T, d):
    """degenerate lengths: one step, fewer steps than an IO group, exactly / just over one group -- every operation"""
    rng = np.random.default_rng(1000 * T + d)
    for tv in (False, True):
        model = U.random_lgssm(rng, tv, d, T)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        check_all(tgp, model, ref.rand(model, *eps), eps)


def test_all_missing_equals_prior(tgp):
    """every observation missing: logpdf is the pure volume compensation (missings.jl:45-53 with innovations of y := 0 under
    variance 1e15), the posterior marginals are the prior marginals"""
    rng = np.random.default_rng(5)
    T, d = 5000, 3
    model = U.random_lgssm(rng, False, d, T)
    y = rng.standard_normal(T)
    missing = np.ones(T, dtype=bool)
    dm = to_device_model(tgp, model)
    yin = np.full(T, np.nan)
    lp = ref.logpdf_missing(model, y, missing)
    # the value is ~0: -T log(2 pi 1e15)/2 from the steps cancels against the compensation; compare on the scale of the terms
    assert abs(tgp.logpdf(dm, yin) - lp) <= 1e-12 * T * np.log(2 * np.pi * 1e15) / 2
    Rn = np.full(T, 0.3)
    pm, pv = tgp.posterior_marginals(dm, yin, Rn)
    mm, mC = ref.marginals(ref.replace_observation_noise_cov(model, Rn))
    np.testing.assert_allclose(pm, mm, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(pv, mC, rtol=1e-7, atol=1e-8)


def test_reference_call_chain_runs_the_fused_smoother(tgp):
    """posterior_lti_sde.jl:27-36 unchanged: marginals(replace_observation_noise_cov(posterior(model, ys), S_new)). On this
    backend posterior() is lazy and replace_observation_noise_cov only records S_new, so the chain launches exactly the
    kernels of the fused tgp_posterior_marginals call -- no materialise pass, no T x (2 d^2 + d) transfer."""
    model, y, _ = U.gp_case(("matern52",), ("regular", 0.0, 0.1, 5000), 0.1, seed=11)
    dm = to_device_model(tgp, model)
    hd = dm.handle()
    Rn = np.full(1, 1e-18)

    def kernels(fn):
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        hd.profile_reset()
        out = fn()
        prof = hd.profile()
        hd.set_option(tgp._lib.OPT_PROFILE, 0)
        return out, {k: v["calls"] for k, v in prof.items()}

    (m1, v1), k1 = kernels(lambda: tgp.posterior_marginals(dm, y, Rn))
    (m2, v2), k2 = kernels(lambda: tgp.marginals(tgp.replace_observation_noise_cov(tgp.posterior(dm, y), Rn)))
    assert k1 == k2 and not any("materialise" in k for k in k2), (k1, k2)
    np.testing.assert_array_equal(m1, m2)
    np.testing.assert_array_equal(v1, v2)
    # without a replacement the posterior keeps the prior's noise (missings.jl:35-41 never called)
    pm, pv = ref.marginals(ref.posterior(model, y))
    gm, gv = tgp.marginals(tgp.posterior(dm, y))
    np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(gv, pv, rtol=1e-8, atol=1e-9)
    # anything that looks inside evaluates the reverse-time model once, and the result is a plain LGSSM
    post = tgp.posterior(dm, y)
    assert post.transitions.As.shape == (5000, 3, 3) and post.ordering is tgp.Reverse


@pytest.mark.parametrize("tv", [False, True])
@pytest.mark.parametrize("d", [1, 2, 3, 4, 6, 9])
def test_posterior_of_a_reverse_ordered_model(tgp, tv, d):
    """step_posterior(::Reverse), lgssm.jl:223-228 (the posterior of a posterior, for one): a Forward-ordered LGSSM whose
    transitions come from invert_dynamics(xp, xf, t) and whose x0 is the state after the last step's predict."""
    rng = np.random.default_rng(500 + d + 50 * tv)
    T = 777
    model = U.random_lgssm(rng, tv, d, T, "R", tame_reverse=True)
    y = rng.standard_normal(T)
    post = ref.posterior(model, y)
    dm = to_device_model(tgp, model)
    for chunk in (0, 7):
        dm.handle().set_option(tgp._lib.OPT_CHUNK, chunk)
        dpost = tgp.posterior(dm, y)
        assert dpost.ordering is tgp.Forward
        np.testing.assert_allclose(dpost.transitions.As, post["A"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.transitions.as_, post["a"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.transitions.Qs, post["Q"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.x0.m, post["x0m"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.x0.P, post["x0P"], rtol=1e-8, atol=1e-9)
    # the reference's chain on it: marginals(replace_observation_noise_cov(posterior(reverse_model, y), R)) (evaluated route), against the
    # ORACLE's posterior model (tame_reverse: a Reverse model whose posterior transitions are contractive, nothing overflows)
    Rn = rng.random(T) * 0.1
    dpost = tgp.posterior(dm, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    assert np.all(np.isfinite(pm)) and np.all(np.isfinite(pv)) and np.abs(pm).max() < 1e3
    gm, gv = tgp.marginals(tgp.replace_observation_noise_cov(dpost, Rn))
    np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(gv, pv, rtol=1e-8, atol=1e-8)


def test_hip_graph_replay_of_repeated_calls(tgp):
    """TGP_OPT_GRAPH: the second call with the same device pointers is recorded (stream capture of the same host code path), later
    ones are replayed by one hipGraphLaunch. Same bits as plain launches; new data behind the same pointers is picked up; any
    other entry point in between drops the recording."""
    import torch
    from temporalgps_jl_amd import _lib
    rng = np.random.default_rng(321)
    T, d = 10_000, 2
    model = U.random_lgssm(rng, False, d, T)
    y_host = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    plain = to_device_model(tgp, model)
    plain.handle_options[_lib.OPT_GRAPH] = 0
    plain.handle_options[_lib.OPT_STEADY] = 1       # (the recorded chain is the general engine's)
    dm = to_device_model(tgp, model)
    dm.handle_options[_lib.OPT_GRAPH] = 1
    dm.handle_options[_lib.OPT_STEADY] = 1
    y = torch.as_tensor(y_host, device="cuda:0")
    Rn = torch.full((1,), 0.07, dtype=torch.float64, device="cuda:0")
    hd = dm.handle()
    base = tgp.logpdf_and_posterior_marginals(plain, y, Rn)
    assert plain.handle().lib.tgp_graph_replays(plain.handle().h) == 0
    out = None
    for it in range(5):
        res = tgp.logpdf_and_posterior_marginals(dm, y, Rn, out=out)
        out = res[1:]
        assert res[0] == base[0]
        assert torch.equal(res[1], base[1]) and torch.equal(res[2], base[2])
    assert hd.lib.tgp_graph_replays(hd.h) == 3            # call 1 plain, call 2 recorded + launched, calls 3..5 replayed
    # new observations behind the same pointer
    y2 = torch.as_tensor(y_host[::-1].copy(), device="cuda:0")
    base2 = tgp.logpdf_and_posterior_marginals(plain, y2, Rn)
    y.copy_(y2)
    res = tgp.logpdf_and_posterior_marginals(dm, y, Rn, out=out)
    assert hd.lib.tgp_graph_replays(hd.h) == 4
    assert res[0] == base2[0] and torch.equal(res[1], base2[1]) and torch.equal(res[2], base2[2])
    # logpdf has its own recording
    lps = [tgp.logpdf(dm, y) for _ in range(4)]
    assert all(v == lps[0] for v in lps) and abs(lps[0] - base2[0]) <= 1e-12 * abs(base2[0])
    assert hd.lib.tgp_graph_replays(hd.h) == 6
    # another entry point in between: the smoother's recording is not replayed blindly
    tgp._filter(dm, y)
    res = tgp.logpdf_and_posterior_marginals(dm, y, Rn, out=out)
    assert hd.lib.tgp_graph_replays(hd.h) == 6
    assert res[0] == base2[0] and torch.equal(res[1], base2[1])
    # missing data + per-step noise, recorded and replayed
    mk = torch.as_tensor((rng.random(T) < 0.1).astype(np.uint8), device="cuda:0")
    yn = (y, mk)                # device observations carry their missing mask as a second tensor
    Rt = torch.as_tensor(rng.uniform(0.01, 0.1, T), device="cuda:0")
    b3 = tgp.logpdf_and_posterior_marginals(plain, yn, Rt)
    out = None
    for it in range(4):
        res = tgp.logpdf_and_posterior_marginals(dm, yn, Rt, out=out)
        out = res[1:]
        assert res[0] == b3[0] and torch.equal(res[1], b3[1]) and torch.equal(res[2], b3[2])


@pytest.mark.parametrize("d_case", [0, 2, 3, 4, 5])
def test_pass1_with_shared_matrix_parts_is_bit_identical(tgp, d_case):
    """TGP_OPT_SHARED_PARTS (default on): Forward LTI model, one noise variance, no missing data -- the matrix parts of every
    chunk's filter element come from one table, pass 1 runs only the vector half of the recursion. Same bits as the general
    pass 1 (the same operations in the same order), for ragged last chunks and several chunk lengths; models that do not qualify
    (missing data, per-step noise) keep the general pass."""
    k, t, s2 = GP_CASES[d_case]
    model, y, _ = U.gp_case(k, t, s2, seed=40 + d_case)
    T = model["T"]
    Rn = np.full(1, 0.05)
    for chunk in (0, 7, 64):
        outs = []
        for opt in (0, 1):
            dm = to_device_model(tgp, model)
            dm.handle_options[tgp._lib.OPT_SHARED_PARTS] = 2 * opt          # 2: table built in line on the first call
            dm.handle_options[tgp._lib.OPT_STEADY] = 1                      # pass 1 of the general engine is what is under test
            hd = dm.handle()
            if chunk:
                hd.set_option(tgp._lib.OPT_CHUNK, chunk)
            hd.set_option(tgp._lib.OPT_PROFILE, 1)
            hd.profile_reset()
            lp = tgp.logpdf(dm, y)
            names = set(hd.profile())
            hd.set_option(tgp._lib.OPT_PROFILE, 0)
            d = len(model["x0m"])
            if d <= 6:
                assert ("k_reduce_filter<lti,shared parts>" in names) == bool(opt), names
            m, P = tgp._filter(dm, y)
            pm, pv = tgp.posterior_marginals(dm, y, Rn)
            ym = y.copy()
            ym[::9] = np.nan
            outs.append((lp, m, P, pm, pv, tgp.logpdf(dm, ym), tgp.posterior_marginals(dm, y, np.full(T, 0.05))[0]))
        for a, b in zip(outs[0], outs[1]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    lp_ref = ref.logpdf(model, y)
    assert abs(outs[1][0] - lp_ref) <= 1e-10 * abs(lp_ref)


def test_shared_parts_table_is_built_off_the_critical_path(tgp):
    """Default policy of TGP_OPT_SHARED_PARTS: the first call on a bound model runs the general pass 1, the second launches the
    table build on a side stream (and still runs the general pass), later calls use the table -- with identical results."""
    import time as _time
    model, y, _ = U.gp_case(("matern52",), ("regular", 0.0, 0.1, 4000), 0.1, seed=77)
    dm = to_device_model(tgp, model)
    dm.handle_options[tgp._lib.OPT_STEADY] = 1        # the general engine
    hd = dm.handle()
    seen, vals = [], []
    for it in range(5):
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        hd.profile_reset()
        vals.append(tgp.logpdf(dm, y))
        seen.append(set(hd.profile()))
        hd.set_option(tgp._lib.OPT_PROFILE, 0)
        _time.sleep(0.05)          # (the side-stream build takes ~1 ms)
    assert "k_reduce_filter<lti>" in seen[0] and "k_reduce_filter<lti>" in seen[1]
    assert "k_reduce_filter<lti,shared parts>" in seen[-1]
    assert all(v == vals[0] for v in vals)


@pytest.mark.parametrize("spec,dt", [(("matern12",), 0.1), (("matern32",), 0.1), (("matern52",), 0.1), (("matern52",), 0.02)])
@pytest.mark.parametrize("chunk", [8, 40, 153])
def test_stationary_covariance_steps_change_no_bit(tgp, spec, dt, chunk):
    """TGP_OPT_STEADY (default on): once a chunk's covariance repeats with period 2 bit for bit, passes 2 and 3 keep only the
    mean half of their steps. Same bits with the option on and off -- logpdf, filtering distributions, posterior marginals with a
    shared and a per-step new noise -- and the mean-only form is really taken (tgp_steady_steps)."""
    import ctypes
    T = 60000
    model, y, _ = U.gp_case(spec, ("regular", 0.0, dt, T), 0.1, seed=31)
    res = {}
    for on in (0, 1):
        dm = to_device_model(tgp, model)
        hd = dm.handle()
        hd.set_option(tgp._lib.OPT_STEADY, on)
        if chunk is not None:
            hd.set_option(tgp._lib.OPT_CHUNK, chunk)
        lp = tgp.logpdf(dm, y)
        m, P = tgp._filter(dm, y)
        lp2, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18]))
        fast, total = ctypes.c_int64(0), ctypes.c_int64(0)
        hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(fast), ctypes.byref(total)))
        Rn = np.random.default_rng(5).random(T) + 0.05
        mean2, var2 = tgp.posterior_marginals(dm, y, Rn)
        res[on] = (lp, m, P, lp2, mean, var, mean2, var2, fast.value, total.value)
    a, b = res[0], res[1]
    assert a[0] == b[0] and a[3] == b[3]
    for i in (1, 2, 4, 5, 6, 7):
        assert np.array_equal(a[i], b[i]), i
    assert a[8] == 0 and b[9] == T
    if chunk == 153 or (chunk == 40 and dt == 0.1):
        # every chunk but the ones inside the filter's initial transient settles within ~15 steps at dt = 0.1, ~30 at dt = 0.02
        assert b[8] > 0.5 * T, (b[8], T)
    lp_ref = sk.logpdf(model, y)
    assert abs(b[0] - lp_ref) <= 1e-10 * abs(lp_ref)
    mean_ref, var_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
    np.testing.assert_allclose(b[4], mean_ref, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(b[5], var_ref, rtol=1e-8, atol=1e-9)


def test_stationary_covariance_steps_are_left_alone_where_they_do_not_apply(tgp):
    """Missing data, per-step noise and per-step models change the map from step to step: tgp_steady_steps reports 0 of T."""
    import ctypes
    T = 20000
    model, y, _ = U.gp_case(("matern52",), ("regular", 0.0, 0.1, T), 0.1, seed=32)
    ym = y.copy()
    ym[[7, 5000, 5001, 19999]] = np.nan
    dm = to_device_model(tgp, model)
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_CHUNK, 64)
    tgp.posterior_marginals(dm, ym, np.array([0.1]))
    fast, total = ctypes.c_int64(-1), ctypes.c_int64(-1)
    hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(fast), ctypes.byref(total)))
    assert fast.value == 0 and total.value == T
    tgp.posterior_marginals(dm, y, np.array([0.1]))
    hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(fast), ctypes.byref(total)))
    assert fast.value > 0.5 * T
    noisy = dict(model, R=np.full(T, 0.1))      # the same noise, but per step: general handling
    dn = to_device_model(tgp, noisy)
    dn.handle().set_option(tgp._lib.OPT_CHUNK, 64)
    tgp.posterior_marginals(dn, y, np.array([0.1]))
    hd2 = dn.handle()
    hd2.check(hd2.lib.tgp_steady_steps(hd2.h, ctypes.byref(fast), ctypes.byref(total)))
    assert fast.value == 0


def test_stationary_covariance_build_is_dropped_where_chunk_0_settles_late(tgp):
    """Kernel choice of the posterior path (tgp_api.hip, `steady_pays`): a pass takes as long as its slowest wave -- the one that holds
    chunk 0, which starts from x0 and settles last. The first call on a bound model reports where chunk 0 switched; if that is
    beyond half of the chunk (a filter that needs hundreds of steps to converge: dt = 0.002 here) later calls run the plain build.
    Same bits either way; a quickly converging model (dt = 0.1) keeps the mean-only steps."""
    import ctypes
    T = 40000
    for dt, keeps in ((0.002, False), (0.1, True)):
        model, y, _ = U.gp_case(("matern52",), ("regular", 0.0, dt, T), 0.1, seed=33)
        dm = to_device_model(tgp, model)
        dm.handle_options[tgp._lib.OPT_STEADY] = 1    # the general engine's per-chunk stationary steps
        hd = dm.handle()
        hd.set_option(tgp._lib.OPT_CHUNK, 153)
        outs, fast = [], []
        for _ in range(3):
            outs.append(tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18])))
            f, t = ctypes.c_int64(0), ctypes.c_int64(0)
            hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(f), ctypes.byref(t)))
            fast.append(f.value)
        assert fast[0] > 0                                    # the first call always tries
        assert (fast[1] > 0) == keeps and (fast[2] > 0) == keeps, (dt, fast)
        for o in outs[1:]:
            assert o[0] == outs[0][0] and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
        hd.set_option(tgp._lib.OPT_CHUNK, 100)                # another chunk length: decided afresh
        tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18]))
        f, t = ctypes.c_int64(0), ctypes.c_int64(0)
        hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(f), ctypes.byref(t)))
        assert f.value > 0


def test_large_pageable_host_arrays_give_the_results_of_device_resident_data(tgp):
    """y in / (mean, var) and filter states out as PAGEABLE host arrays of tens of MB (what a Julia Vector{Float64} caller hands over):
    bitwise the results of the same calls on device-resident data."""
    import torch
    from temporalgps_jl_amd import lti_sde
    T = 3_100_003                      # 23.65 MiB of observations: two full chunks + a ragged one
    model = lti_sde.build_lgssm(lti_sde.Matern32Kernel(), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
    y = np.random.default_rng(31).standard_normal(T)
    yd = torch.as_tensor(y, device="cuda:0")
    rn = np.array([0.05])
    assert tgp.logpdf(model, y) == tgp.logpdf(model, yd)
    mean, var = tgp.posterior_marginals(model, y, rn)
    md, vd = tgp.posterior_marginals(model, yd, torch.as_tensor(rn, device="cuda:0"))
    assert np.array_equal(mean, md.cpu().numpy()) and np.array_equal(var, vd.cpu().numpy())
    mf, Pf = tgp._filter(model, y)
    mfd, Pfd = tgp._filter(model, yd)
    assert np.array_equal(mf, mfd.cpu().numpy()) and np.array_equal(Pf, Pfd.cpu().numpy())


def test_nan_observations_in_a_large_host_series_are_found_without_scanning_every_call(tgp):
    """NaN == missing for host arrays. Large scalar-output series are not scanned on the host before the call (the scan costs more than
    the PCIe transfer): the NaN comes back as a NaN log-likelihood and the call is repeated with the mask -- same results as an explicit
    mask, for the stationary-gain engine's models (LTI) and the general engine's (per-step) alike."""
    from temporalgps_jl_amd import lgssm as L, lti_sde
    T = 200_000
    assert T >= L._LAZY_NAN_MIN
    rng = np.random.default_rng(41)
    y = rng.standard_normal(T)
    miss = rng.random(T) < 0.01
    yn = y.copy()
    yn[miss] = np.nan
    rn = np.array([0.05])
    for per_step in (False, True):
        model = lti_sde.build_lgssm(lti_sde.Matern32Kernel(), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1, force_per_step=per_step)
        want = tgp.logpdf(model, (y, miss))
        assert np.isfinite(want)
        assert tgp.logpdf(model, yn) == want
        m0, v0 = tgp.posterior_marginals(model, (y, miss), rn)
        m1, v1 = tgp.posterior_marginals(model, yn, rn)
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
        lp, m2, v2 = tgp.logpdf_and_posterior_marginals(model, yn, rn)
        assert lp == want and np.array_equal(m0, m2)
        # no NaN: one call, no mask (an explicit all-false mask takes the LTI model off the stationary-gain engine: equal to rounding)
        assert abs(tgp.logpdf(model, y) - tgp.logpdf(model, (y, np.zeros(T, dtype=bool)))) <= 1e-12 * abs(want)


@pytest.mark.parametrize("kernel,d", [(("matern32",), 2), (("matern52",), 3), (("sum", ("matern32",), ("matern52",)), 5)])
@pytest.mark.parametrize("noise", ["shared", "per-step"])
def test_logpdf_of_an_unevaluated_posterior_needs_no_posterior(tgp, kernel, d, noise):
    """posterior_lti_sde.jl:62-78's last line, logpdf(replace_observation_noise_cov(posterior(model, ys), S_new), ys_new): on this backend two
    logpdf calls of the PRIOR through the pair statistic (tgp_pair_statistic for device-resident series) -- held against the
    oracle's literal chain (posterior evaluated, lgssm.jl:193-221, then filtered, :147-151) and against the product's own evaluated route."""
    import torch
    rng = np.random.default_rng(5)
    T = 3000
    s2 = 0.3 if noise == "shared" else rng.random(T) * 0.3 + 0.1
    model, y, _ = U.gp_case(kernel, ("regular", 0.0, 0.1, T), s2, seed=3)
    assert len(model["x0m"]) == d
    dm = to_device_model(tgp, model)
    y_new = y + 0.4 * rng.standard_normal(T)
    R_new = np.full(1, 0.2) if noise == "shared" else rng.random(T) * 0.2 + 0.05
    want = ref.logpdf(ref.replace_observation_noise_cov(ref.posterior(model, y), R_new), y_new)
    chain = lambda yy, yn, Rn: tgp.logpdf(tgp.replace_observation_noise_cov(tgp.posterior(dm, yy), Rn), yn)
    post = tgp.replace_observation_noise_cov(tgp.posterior(dm, y), R_new)
    got = tgp.logpdf(post, y_new)
    assert post._model is None                                     # no reverse-time model evaluated
    assert abs(got - want) <= 1e-8 * abs(want)
    evaluated = tgp.logpdf(tgp.replace_observation_noise_cov(tgp.posterior(dm, y), R_new).materialise(), y_new)
    assert abs(got - evaluated) <= 1e-8 * abs(want)
    # device-resident series (and a device-resident per-step noise): the statistic by tgp_pair_statistic
    dev = torch.device("cuda:0")
    yd, ynd = torch.as_tensor(y, device=dev), torch.as_tensor(y_new, device=dev)
    Rnd = R_new if noise == "shared" else torch.as_tensor(R_new, device=dev)
    got_dev = chain(yd, ynd, Rnd)
    assert abs(got_dev - got) <= 1e-12 * abs(want)
    # missing entries on either side, on both, host (NaN) and device (masks)
    ym, ynm = y.copy(), y_new.copy()
    ym[[5, 700, 701]] = np.nan
    ynm[[9, 700, 2999]] = np.nan
    miss, miss_new = np.isnan(ym), np.isnan(ynm)
    post_o = ref.replace_observation_noise_cov(ref.posterior_missing(model, np.nan_to_num(ym), miss), R_new)
    want_m = ref.logpdf_missing(post_o, np.nan_to_num(ynm), miss_new)
    got_m = chain(ym, ynm, R_new)
    assert abs(got_m - want_m) <= 1e-8 * abs(want_m)
    got_md = chain((yd, torch.as_tensor(miss, device=dev)), (ynd, torch.as_tensor(miss_new, device=dev)), Rnd)
    assert abs(got_md - got_m) <= 1e-12 * abs(want_m)


def test_pair_statistic_abi_argument_checks(tgp):
    import ctypes
    import torch
    dm = to_device_model(tgp, U.gp_case(("matern32",), ("regular", 0.0, 0.1, 50), 0.1, seed=1)[0])
    hd = dm.handle()
    dev = torch.device("cuda:0")
    y = torch.zeros(50, dtype=torch.float64, device=dev)
    out, pair = torch.empty_like(y), ctypes.c_double()
    ptr, R = tgp._lib.ptr, np.array([0.5])
    call = lambda *a: hd.lib.tgp_pair_statistic(hd.h, *a)
    assert call(50, ptr(y), None, ptr(R), 7, ptr(y), None, ptr(R), 1, ptr(out), None, None, ctypes.byref(pair)) == tgp._lib.EINVAL     # 7 variances for 50 steps
    Rt = torch.full((50,), 0.5, dtype=torch.float64, device=dev)
    assert call(50, ptr(y), None, ptr(Rt), 50, ptr(y), None, ptr(R), 1, ptr(out), None, None, ctypes.byref(pair)) == tgp._lib.EINVAL   # per-step noise without Rbar
    assert call(0, ptr(y), None, ptr(R), 1, ptr(y), None, ptr(R), 1, ptr(out), None, None, ctypes.byref(pair)) == 0 and pair.value == 0.0
    assert call(50, ptr(y), None, ptr(R), 1, ptr(y), None, ptr(R), 1, ptr(out), None, None, ctypes.byref(pair)) == 0
    torch.cuda.synchronize()
    assert abs(pair.value + 25 * np.log(2 * np.pi)) < 1e-12 and float(out.abs().max()) == 0.0


@pytest.mark.parametrize("kernel", [("matern52",), ("sum", ("matern52",), ("matern52",)), ("sum", ("matern32",), ("matern12",))])
def test_logpdf_with_another_noise_variance_on_the_bound_handle(tgp, kernel):
    """tgp_logpdf_noise: logpdf of the bound LTI model with ONE other noise variance == tgp_logpdf of the model bound with it (oracle beside
    both), modal and dense-powers one-launch paths; the bound model's own calls are untouched by it; models off those paths are refused."""
    import ctypes
    T = 6000
    model, y, _ = U.gp_case(kernel, ("regular", 0.0, 0.1, T), 0.3, seed=8)
    dm = to_device_model(tgp, model)
    hd = dm.handle()
    before = tgp.logpdf(dm, y)
    for R in (0.07, 2.5, 1e-6):
        other = dict(model, R=np.array([R]))
        want = ref.logpdf(other, y)
        out = ctypes.c_double()
        rc = hd.lib.tgp_logpdf_noise(hd.h, tgp._lib.ptr(y), 0, R, ctypes.byref(out))
        if rc == tgp._lib.EUNSUPPORTED and R < 1e-3:       # (a nearly noise-free model may be declined by every one-launch plan: the caller binds it)
            assert tgp.logpdf(dm, y) == before
            continue
        assert rc == 0
        assert abs(out.value - want) <= 1e-9 * abs(want)
        assert abs(out.value - tgp.logpdf(to_device_model(tgp, other), y)) <= 1e-12 * abs(want)
        assert tgp.logpdf(dm, y) == before
    out = ctypes.c_double()
    assert hd.lib.tgp_logpdf_noise(hd.h, tgp._lib.ptr(y), 0, -1.0, ctypes.byref(out)) == tgp._lib.EINVAL
    per_step = to_device_model(tgp, dict(model, R=np.full(T, 0.3)))
    hp = per_step.handle()
    assert hp.lib.tgp_logpdf_noise(hp.h, tgp._lib.ptr(y), 0, 0.1, ctypes.byref(out)) == tgp._lib.EUNSUPPORTED
