"""GPU tier: the stationary-gain scan engine (csrc/tgp_steady.hip, TGP_OPT_STEADY = 2, the default for Forward LTI models with one
noise variance, scalar observations and no missing data -- the reference's Fill layout, lti_sde.jl:148-160) against the oracle's
sequential restatement of lgssm.jl:99-238, through the C ABI.  Tolerances as everywhere: logpdf 1e-10 relative, marginals 1e-8."""
import ctypes

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


def device_model(tgp, model, steady=2):      # (TGP_OPT_STEADY = 2: this file tests tgp_steady.hip; the one-launch form of round 4 has tests/test_gpu_modal.py)
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])
    if steady is not None:
        dm.handle_options[tgp._lib.OPT_STEADY] = steady
    return dm


def served(dm):
    """steps of the last call that ran with the stationary gains (0: the general engine served it)"""
    hd = dm.handle()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(a), ctypes.byref(b)))
    return a.value


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
    return sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))


KERNELS = {
    1: ("matern12",),
    2: ("matern32",),
    3: ("matern52",),
    4: ("sum", ("matern52",), ("matern12",)),
    5: ("sum", ("matern52",), ("matern32",)),
    6: ("sum", ("matern52",), ("matern52",)),
    7: ("sum", ("matern52",), ("matern32",), ("matern32",)),
    8: ("sum", ("matern52",), ("matern52",), ("matern32",)),
}


@pytest.mark.parametrize("d", sorted(KERNELS))
def test_every_state_dimension_against_the_oracle(tgp, d):
    """logpdf, posterior marginals and the combined call for d = 1..8 (one kernel template each); the engine, not the fallback, ran"""
    T = 6000 + 37 * d
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, d)
    lp_ref = sk.logpdf(model, y)
    Rn = np.array([0.02])
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    dm = device_model(tgp, model)
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    assert "k_steady_apply<logpdf>" in names and not any(n.startswith("k_reduce_filter") for n in names), names
    assert served(dm) > T - 300
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
    assert "k_steady_apply<posterior>" in names and "k_smooth<lti>" not in names, names
    np.testing.assert_allclose(mean, m_ref, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, v_ref, rtol=1e-8, atol=1e-10)
    lp2, mean2, var2 = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    assert abs(lp2 - lp_ref) <= 1e-10 * abs(lp_ref)
    np.testing.assert_allclose(mean2, m_ref, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var2, v_ref, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("T", [575, 1024, 1025, 4095, 4096, 4097, 4608, 4609, 8191, 8200, 12289, 36864 + 5])
def test_ragged_ends_and_tile_boundaries(tgp, T):
    """series lengths around the tile (512), workgroup (4096) and head boundaries: a ragged last tile / workgroup takes its own coupling
    matrices; the tail of the smoothed variance may reach across several tiles"""
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, T)
    dm = device_model(tgp, model)
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18]))
    assert served(dm) > 0
    lp_ref = sk.logpdf(model, y)
    m_ref, v_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    np.testing.assert_allclose(mean, m_ref, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, v_ref, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("dt", [0.3, 0.03, 0.01, 0.004])
def test_heads_of_different_lengths(tgp, dt):
    """the filter covariance settles after ~20 (dt = 0.3) to ~1000 (dt = 0.004) steps: heads of one to several tiles, per-step gains from
    the tables, general affine scans over the head's lanes"""
    T = 30000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, dt, T), 0.1)
    y = draw(model, 17)
    dm = device_model(tgp, model)
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18]))
    n_head = T - served(dm)
    assert 0 < n_head < 2048, n_head
    lp_ref = sk.logpdf(model, y)
    m_ref, v_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    np.testing.assert_allclose(mean, m_ref, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, v_ref, rtol=1e-8, atol=1e-10)


def test_mean_function_scaled_stretched_kernel_and_per_step_new_noise(tgp):
    """a shared emission offset h != 0 (a mean function makes h a per-step vector in the reference, lti_sde.jl:118-131: that layout keeps
    the general engine; a Fill offset is the LTI case), kernel algebra (:334-373), a per-step R_new (missings.jl:35-41); agreement with
    the general engine"""
    T = 9000
    k = ("scaled", 2.5, ("stretched", 0.7, ("matern32",)))
    model = dict(oc.build_lgssm(k, ("regular", 0.0, 0.1, T), 0.5), h=np.array([1.5]))
    y = draw(model, 5)
    rng = np.random.default_rng(6)
    Rn = 0.1 * (1.0 + rng.random(T))
    dm = device_model(tgp, model)
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    assert served(dm) > 0
    post = ref.posterior(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    lp_ref = ref.logpdf(model, y)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    np.testing.assert_allclose(mean, pm, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, pv, rtol=1e-8, atol=1e-10)
    dg = device_model(tgp, model, steady=1)
    lpg, meang, varg = tgp.logpdf_and_posterior_marginals(dg, y, Rn)
    assert served(dg) >= 0 and abs(lp - lpg) <= 1e-11 * abs(lpg)
    np.testing.assert_allclose(mean, meang, rtol=0, atol=1e-9)
    np.testing.assert_allclose(var, varg, rtol=1e-9, atol=1e-11)


def test_random_lti_models(tgp):
    """test/models/model_test_utils.jl-style random stable LTI models (dense A, Q, non-zero a and h, x0 not the stationary state)"""
    for d in (1, 2, 3, 4, 6):
        rng = np.random.default_rng(100 + d)
        T = 5000
        model = U.random_lgssm(rng, False, d, T)
        y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        dm = device_model(tgp, model)
        Rn = np.array([0.3])
        lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
        assert served(dm) > 0
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        np.testing.assert_allclose(mean, m_ref, rtol=0, atol=1e-8)
        np.testing.assert_allclose(var, v_ref, rtol=1e-8, atol=1e-10)


def test_where_the_engine_does_not_apply_the_general_path_serves(tgp):
    """decided on the device inside the call: a series shorter than head + tail, and a filter that needs more than 2048 steps to settle
    (dt = 0.0005); the handle remembers, the next call does not try again; results are the oracle's either way"""
    for k, dt, T in ((("matern52",), 0.1, 560), (("matern52",), 0.0005, 20000)):
        model = oc.build_lgssm(k, ("regular", 0.0, dt, T), 0.1)
        y = draw(model, 3)
        dm = device_model(tgp, model)
        (lp, mean, var), names1 = kernels_of(tgp, dm, lambda: tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18])))
        assert served(dm) == 0 or served(dm) < T        # general engine's own count
        assert "k_steady_setup" in names1 and any(n.startswith("k_apply_filter") for n in names1), names1
        _, names2 = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
        assert not any(n.startswith("k_steady") for n in names2), names2
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        np.testing.assert_allclose(mean, m_ref, rtol=0, atol=1e-8)
        np.testing.assert_allclose(var, v_ref, rtol=1e-8, atol=1e-10)


def test_models_outside_the_layout_keep_the_general_engine(tgp):
    """missing data, per-step noise, an explicit chunk length: the general engine, whatever TGP_OPT_STEADY says"""
    T = 5000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 8)
    dm = device_model(tgp, model)
    ym = y.copy()
    ym[::7] = np.nan
    _, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, ym))
    assert not any(n.startswith("k_steady") for n in names), names
    dm.handle().set_option(tgp._lib.OPT_CHUNK, 40)
    _, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    assert not any(n.startswith("k_steady") for n in names), names
    het = dict(model, R=np.full(T, 0.1))
    dh = device_model(tgp, het)
    lp, names = kernels_of(tgp, dh, lambda: tgp.logpdf(dh, y))
    assert not any(n.startswith("k_steady") for n in names), names
    assert abs(lp - sk.logpdf(model, y)) <= 1e-10 * abs(lp)


def test_not_positive_definite_is_reported(tgp):
    """a negative innovation variance (Julia: DomainError from sqrt, lgc.jl:250) -> TGP_ENOTPD, from k_steady_setup"""
    T = 3000
    model = oc.build_lgssm(("matern32",), ("regular", 0.0, 0.1, T), 0.1)
    model = dict(model, R=np.array([-5.0]))
    dm = device_model(tgp, model)
    with pytest.raises(tgp._lib.NotPositiveDefinite):
        tgp.logpdf(dm, np.zeros(T))


def test_device_buffers_and_repeated_calls(tgp):
    """device-resident y / outputs (the bench's call), repeated on one handle with new data behind the same pointers"""
    import torch
    T = 200_000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    dm = device_model(tgp, model)
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    out = None
    yd = torch.empty(T, dtype=torch.float64, device="cuda:0")
    for seed in (1, 2, 3):
        y = draw(model, seed)
        yd.copy_(torch.as_tensor(y))
        res = tgp.logpdf_and_posterior_marginals(dm, yd, Rn, out=out)
        out = res[1:]
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
        assert abs(res[0] - lp_ref) <= 1e-10 * abs(lp_ref)
        np.testing.assert_allclose(res[1].cpu().numpy(), m_ref, rtol=0, atol=1e-8)
        np.testing.assert_allclose(res[2].cpu().numpy(), v_ref, rtol=1e-8, atol=1e-10)


def test_several_carry_slices(tgp):
    """T = 2.2e7 at d = 3: more than 4096 workgroups, the carry kernel walks two slices chained through the slice's end state"""
    import torch
    T = 22_000_003
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 11)
    dm = device_model(tgp, model)
    yd = torch.as_tensor(y, device="cuda:0")
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, yd, Rn)
    assert served(dm) > T - 200
    lp_ref = sk.logpdf(model, y)
    m_ref, v_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    assert float(np.max(np.abs(mean.cpu().numpy() - m_ref))) <= 1e-8
    assert float(np.max(np.abs(var.cpu().numpy() - v_ref))) <= 1e-8


def test_cfg1_exact_size(tgp):
    """BASELINE config 1 at its exact size: Matern-3/2, RegularSpacing(0, 0.1, 10_000), sigma^2 = 0.1 (README example /
    bench/single_output_gps.jl) against the NumPy restatement oracle.lgssm_ref, both engines"""
    T = 10_000
    model = oc.build_lgssm(("matern32",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 1)
    lp_ref = ref.logpdf(model, y)
    post = ref.posterior(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, np.array([1e-18])))
    for steady in (3, 2, 1):      # 3: the default -- the one-launch engine, at this length with the head as scans (what bench.py times for cfg1)
        dm = device_model(tgp, model, steady)
        hd = dm.handle()
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        hd.profile_reset()
        lp = tgp.logpdf(dm, y)
        mean, var = tgp.posterior_marginals(dm, y, np.array([1e-18]))
        lp2, mean2, var2 = tgp.logpdf_and_posterior_marginals(dm, y, np.array([1e-18]))
        names = set(hd.profile())
        hd.set_option(tgp._lib.OPT_PROFILE, 0)
        assert (served(dm) > 9900) == (steady >= 2)
        if steady == 3:
            assert names and all(n.startswith(("k_steady_one", "k_lml_stream", "k_post_stream")) for n in names), names
        elif steady == 2:
            assert not any(n.startswith(("k_steady_one", "k_lml_stream", "k_post_stream")) for n in names) and any(n.startswith("k_steady") for n in names), names
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref) and abs(lp2 - lp_ref) <= 1e-10 * abs(lp_ref)
        for mm, vv in ((mean, var), (mean2, var2)):
            np.testing.assert_allclose(mm, pm, rtol=0, atol=1e-8)
            np.testing.assert_allclose(vv, pv, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("d", [2, 3, 4, 6])
@pytest.mark.parametrize("T", [9 * 512, 9 * 512 + 5, 40_001])
def test_device_pointers_off_the_16_byte_boundary(tgp, d, T):
    """Pass 2 writes a tile's outputs as whole 1 KB rows (transposed through LDS) and pass 1 / 2 read 16-byte pairs -- behind pointers that
    sit on a 16-byte boundary. Views into larger device arrays start wherever the caller's offset puts them: observations, per-step new
    noise and both outputs at an odd multiple of 8 bytes, whole and ragged last tiles."""
    import torch
    k = {2: ("matern32",), 3: ("matern52",), 4: ("sum", ("matern52",), ("matern12",)), 6: ("sum", ("matern52",), ("matern52",))}[d]
    model = oc.build_lgssm(k, ("regular", 0.0, 0.1, T), 0.1)
    rng = np.random.default_rng(d + T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = sk.rand(model, *eps)
    Rn = rng.random(T) * 0.1 + 0.01
    lp_o = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, Rn)
    dm = device_model(tgp, model)
    dev = torch.device("cuda:0")
    for off in (1, 0, 3):
        ybig = torch.zeros(T + 8, dtype=torch.float64, device=dev)
        rbig = torch.zeros(T + 8, dtype=torch.float64, device=dev)
        mbig = torch.full((T + 8,), 7.0, dtype=torch.float64, device=dev)
        vbig = torch.full((T + 8,), 7.0, dtype=torch.float64, device=dev)
        ybig[off:off + T] = torch.as_tensor(y, device=dev)
        rbig[off:off + T] = torch.as_tensor(Rn, device=dev)
        out = (mbig[off:off + T], vbig[off:off + T])
        assert (out[0].data_ptr() % 16 == 8) == (off % 2 == 1)
        lp, gm, gv = tgp.logpdf_and_posterior_marginals(dm, ybig[off:off + T], rbig[off:off + T], out=out)
        assert served(dm) > 0
        assert abs(lp - lp_o) <= 1e-10 * abs(lp_o)
        np.testing.assert_allclose(gm.cpu().numpy(), pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv.cpu().numpy(), pv, rtol=1e-8, atol=1e-9)
        for big in (mbig, vbig):                     # nothing written outside the view
            assert torch.all(big[:off] == 7.0) and torch.all(big[off + T:] == 7.0)
        assert abs(tgp.logpdf(dm, ybig[off:off + T]) - lp_o) <= 1e-10 * abs(lp_o)
