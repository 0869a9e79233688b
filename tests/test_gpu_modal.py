"""GPU tier: the ONE-LAUNCH form of the stationary-gain engine (csrc/tgp_modal.hip + the host plan csrc/tgp_steady_plan.hpp;
TGP_OPT_STEADY = 3, the default for Forward LTI models with one noise variance, scalar observations and no missing data -- the reference's
Fill layout, lti_sde.jl:148-160) against the oracle's sequential restatement of lgssm.jl:99-238, through the C ABI.
Tolerances as everywhere: logpdf 1e-10 relative, marginals 1e-8."""
import ctypes

import numpy as np
import pytest

from oracle import components as oc
from oracle import seq_kalman as sk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def device_model(tgp, model, steady=None):
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])
    if steady is not None:
        dm.handle_options[tgp._lib.OPT_STEADY] = steady
    return dm


def served(dm):
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


# one kernel per state dimension whose stationary closed loop has a well-conditioned modal form (distinct length scales)
KERNELS = {
    1: ("matern12",),
    2: ("matern32",),
    3: ("matern52",),
    4: ("sum", ("matern52",), ("matern12",)),
    5: ("sum", ("matern52",), ("matern32",)),
    6: ("sum", ("matern52",), ("stretched", 0.4, ("matern52",))),
    7: ("sum", ("matern52",), ("stretched", 0.5, ("matern32",)), ("scaled", 0.3, ("matern32",))),
    8: ("sum", ("matern52",), ("stretched", 2.0, ("matern52",)), ("stretched", 0.5, ("matern32",))),
}


def check(tgp, model, y, Rn, T, expect_one=True, tol_m=1e-8):
    lp_ref = sk.logpdf(model, y)
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    dm = device_model(tgp, model)
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    if expect_one:
        assert any(n.startswith("k_lml_stream") or (n.startswith("k_steady_one") and "logpdf" in n) for n in names) and len(names) == 1, names
        assert served(dm) > T - 700
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (lp, lp_ref)
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
    if expect_one:
        assert any(n.startswith("k_post_stream") or (n.startswith("k_steady_one") and "posterior" in n) for n in names) and len(names) == 1, names
    assert np.max(np.abs(mean - m_ref)) <= tol_m, np.max(np.abs(mean - m_ref))
    assert np.max(np.abs(var - v_ref)) <= tol_m, np.max(np.abs(var - v_ref))
    lp2, mean2, var2 = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    assert abs(lp2 - lp_ref) <= 1e-10 * abs(lp_ref)
    assert np.array_equal(mean2, mean) and np.array_equal(var2, var)
    return dm


@pytest.mark.parametrize("d", sorted(KERNELS))
def test_every_state_dimension_against_the_oracle(tgp, d):
    T = 21000 + 37 * d
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, d)
    check(tgp, model, y, np.array([0.02]), T)


@pytest.mark.parametrize("T", [401, 512, 513, 1000, 3583, 3584, 3585, 3776, 3777, 4096, 4097, 7551, 7552, 7553, 8191, 12345])
def test_series_lengths_around_every_boundary(tgp, T):
    """whole and ragged last tiles, one / two / several workgroups (the core of a workgroup is 8 x 512 - 2 halo steps)"""
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, T)
    check(tgp, model, y, np.array([0.05]), T)


@pytest.mark.parametrize("dt,s2", [(0.03, 0.5), (0.3, 0.01), (1.0, 1e-3), (0.1, 3.0), (0.02, 0.05)])
def test_spacings_and_noise_levels(tgp, dt, s2):
    """halo lengths from one tile's worth to several (slow mixing: 16 waves per workgroup), heads of 20 to 200 steps"""
    T = 60011
    model = oc.build_lgssm(("sum", ("matern52",), ("matern32",)), ("regular", 0.0, dt, T), s2)
    y = draw(model, 3)
    check(tgp, model, y, np.array([0.1]), T)


def test_per_step_new_noise_and_offsets(tgp):
    """R_new per step; a model with a transition offset a and an emission offset h"""
    T = 17001
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.2)
    rng = np.random.default_rng(5)
    model["a"] = np.broadcast_to(0.1 * rng.standard_normal(3), np.asarray(model["a"]).shape).copy()
    model["h"] = np.broadcast_to(np.array(0.7), np.asarray(model["h"]).shape).copy()
    y = draw(model, 9)
    Rn = 0.01 + rng.random(T)
    check(tgp, model, y, Rn, T)


def test_device_pointers_off_the_16_byte_boundary(tgp):
    import torch
    T = 20000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 2)
    Rn = np.array([0.03])
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    lp_ref = sk.logpdf(model, y)
    dm = device_model(tgp, model)
    buf = torch.zeros(T + 1, dtype=torch.float64, device="cuda")
    buf[1:] = torch.from_numpy(y).cuda()
    yo = buf[1:]                                     # 8 bytes past a 16-byte boundary
    outm = torch.zeros(T + 1, dtype=torch.float64, device="cuda")
    outv = torch.zeros(T + 1, dtype=torch.float64, device="cuda")
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, yo, torch.tensor(Rn, device="cuda"), out=(outm[1:], outv[1:]))
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    assert np.max(np.abs(mean.cpu().numpy() - m_ref)) <= 1e-8 and np.max(np.abs(var.cpu().numpy() - v_ref)) <= 1e-8
    assert float(outm[0]) == 0.0 and float(outv[0]) == 0.0


DECLINED = {      # no well-conditioned modal form: summands with ONE length scale (a defective closed loop)
    4: ("sum", ("matern32",), ("matern32",)),      # (two Matern-1/2 with one length scale: a double eigenvalue, but diagonalisable -- the modal plan takes it)
    6: ("sum", ("matern52",), ("matern52",)),      # SURVEY 8d's own cfg3 at d = 6
    7: ("sum", ("matern32",), ("matern32",), ("matern52",)),
    8: ("sum", ("matern52",), ("matern52",), ("matern32",)),
}


@pytest.mark.parametrize("d", sorted(DECLINED))
def test_models_the_plan_declines_run_in_one_launch_on_dense_powers(tgp, d):
    """a sum of two identical kernels has no well-conditioned modal form: logpdf by k_smooth_one's forward half, posterior marginals (+ logpdf) by
    k_smooth_one -- both recursions on DENSE powers (closed loop forwards, settled reverse-time transition backwards), ONE launch each
    (lgssm.jl:111-115, :215-238)"""
    T = 9000
    model = oc.build_lgssm(DECLINED[d], ("regular", 0.0, 0.1, T), 0.1)
    assert len(model["x0m"]) == d
    y = draw(model, d)
    dm = device_model(tgp, model)
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    assert names == {"k_smooth_one<logpdf>"}, names      # (its forward half alone)
    assert served(dm) > T - 700
    lp_ref = sk.logpdf(model, y)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    Rn = np.array([0.2])
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
    assert names == {"k_smooth_one<posterior>"}, names
    assert served(dm) > T - 700
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8
    lp2, mean2, var2 = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    assert abs(lp2 - lp_ref) <= 1e-10 * abs(lp_ref)
    assert np.array_equal(mean2, mean) and np.array_equal(var2, var)
    assert abs(tgp.logpdf(dm, y) - lp_ref) <= 1e-10 * abs(lp_ref)      # (and again behind a posterior call)
    # TGP_OPT_STEADY = 2 keeps the five-launch engine for the same model
    d2 = device_model(tgp, model, steady=2)
    (mean5, var5), names = kernels_of(tgp, d2, lambda: tgp.posterior_marginals(d2, y, Rn))
    assert "k_steady_apply<posterior>" in names and not any(n.startswith("k_smooth_one") for n in names), names
    assert np.max(np.abs(mean5 - mean)) <= 1e-9 and np.max(np.abs(var5 - var)) <= 1e-9


@pytest.mark.parametrize("T", [700, 1999, 3584, 3585, 4096, 4097, 7000, 8191, 12289, 40000, 300007])
def test_dense_powers_smoother_lengths_around_every_boundary(tgp, T):
    """series ending inside / at the edge of a tile, a span, the variance transient; per-step new noise; device-resident inputs and outputs"""
    import torch
    model = oc.build_lgssm(DECLINED[6], ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, T)
    rng = np.random.default_rng(T)
    Rn = rng.random(T) * 0.1
    lp_ref = sk.logpdf(model, y)
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    dm = device_model(tgp, model)
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
    assert names == {"k_smooth_one<posterior>"}, names
    assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8
    # device pointers (off the 16-byte boundary as well)
    hd = dm.handle()
    L = tgp._lib
    for off in (0, 1):
        yd = torch.zeros(T + 2, dtype=torch.float64, device="cuda")
        yd[off:off + T] = torch.from_numpy(y).cuda()
        rd = torch.from_numpy(Rn).cuda()
        md = torch.zeros(T + 2, dtype=torch.float64, device="cuda")
        vd = torch.zeros(T + 2, dtype=torch.float64, device="cuda")
        out = ctypes.c_double()
        ptr = lambda t, o=0: ctypes.c_void_p(t.data_ptr() + 8 * o)
        torch.cuda.synchronize()      # (device inputs are the caller's to complete: the library's streams do not wait for torch's)
        hd.check(hd.lib.tgp_logpdf_and_posterior_marginals(hd.h, ptr(yd, off), None, ptr(rd), L.IN_DEVICE | L.OUT_DEVICE, ctypes.byref(out),
                                                          ptr(md, off), ptr(vd, off)))
        torch.cuda.synchronize()
        assert abs(out.value - lp_ref) <= 1e-10 * abs(lp_ref)
        assert np.max(np.abs(md[off:off + T].cpu().numpy() - m_ref)) <= 1e-8
        assert np.max(np.abs(vd[off:off + T].cpu().numpy() - v_ref)) <= 1e-8
        assert float(md[off + T:].abs().max()) == 0.0 and float(vd[off + T:].abs().max()) == 0.0      # (nothing written behind the series)


def test_dense_powers_smoother_steps_aside(tgp):
    """a series too short for head + transient, and an explicit chunk length: the older engines serve the call"""
    model = oc.build_lgssm(DECLINED[6], ("regular", 0.0, 0.1, 90), 0.1)
    y = draw(model, 5)
    dm = device_model(tgp, model)
    Rn = np.array([0.1])
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
    assert not any(n.startswith("k_smooth_one") for n in names), names
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8


def test_option_2_keeps_the_five_launch_engine(tgp):
    T = 9000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 1)
    dm = device_model(tgp, model, steady=2)
    lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
    assert "k_steady_apply<logpdf>" in names and not any(n.startswith(("k_steady_one", "k_lml_stream", "k_post_stream")) for n in names), names
    d3 = device_model(tgp, model)
    lp3 = tgp.logpdf(d3, y)
    assert abs(lp - lp3) <= 1e-11 * abs(lp)


def test_nan_observation_gives_nan(tgp):
    """(the lazy missing-data path of the Python mirror relies on a NaN observation surfacing as a NaN log-likelihood)"""
    T = 9000
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 1)
    y[5000] = np.nan
    import torch
    dm = device_model(tgp, model)
    hd = dm.handle()
    yd = torch.from_numpy(y).cuda()
    out = ctypes.c_double()
    rc = hd.lib.tgp_logpdf(hd.h, ctypes.c_void_p(yd.data_ptr()), None, tgp._lib.IN_DEVICE, ctypes.byref(out))
    assert rc != 0 or np.isnan(out.value)


@pytest.mark.parametrize("ordering", ["F", "R"])
@pytest.mark.parametrize("d", [1, 3, 6, 8])
def test_prior_marginals_of_an_lti_model_are_a_head_and_a_constant(tgp, d, ordering):
    """tgp_marginals of an LTI model (lgssm.jl:99-115): the host runs the data-free recursion to its fixed point, the device writes the head and
    the constant (k_fill_marginals).  Random LTI models (x0 is NOT the stationary distribution: a real head), both orderings; a GP prior."""
    from oracle import lgssm_ref as ref
    from tests import _util as U
    rng = np.random.default_rng(100 * d + (ordering == "R"))
    T = 3000
    model = U.random_lgssm(rng, False, d, T, ordering)
    tr = tgp.GaussMarkovModel(tgp.Forward if ordering == "F" else tgp.Reverse, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
    (gm, gv), names = kernels_of(tgp, dm, lambda: tgp.marginals(dm))
    assert names == {"k_fill_marginals<lti>"}, names
    qm, qv = ref.marginals(model)
    np.testing.assert_allclose(gm, qm, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gv, qv, rtol=1e-10, atol=1e-10)
    if ordering == "F" and d in (3, 6):
        gp = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, 50_000), 0.1)
        dg = device_model(tgp, gp)
        (pm, pv), names = kernels_of(tgp, dg, lambda: tgp.marginals(dg))
        assert names == {"k_fill_marginals<lti>"}, names
        rm, rv = ref.marginals(dict(gp, T=2000))
        assert np.max(np.abs(pm[:2000] - rm)) <= 1e-10 and np.max(np.abs(pv[:2000] - rv)) <= 1e-10 and np.ptp(pv[100:]) <= 1e-12


def test_both_forms_of_the_head_for_small_state_dimensions():
    """d <= 4 and at most 512 workgroups: the head runs as scans over its steps (the kernel's second instantiation); longer series keep the
    sequential head.  Both against the oracle on the same short series -- the sequential form through TGP_MODAL_HEAD_SCANS=0 in a
    process of its own (the library reads the variable once) --, and a long head (slow covariance: several 64-step tiles of scans)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
import temporalgps_jl_amd as tgp
from oracle import components as oc
from tests.test_gpu_modal import KERNELS, check, draw
for d in (1, 2, 3, 4):
    for T, dt, noise in ((3000, 0.1, 0.1), (20011, 0.02, 0.003)):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), noise)
        y = draw(model, 400 + d)
        check(tgp, model, y, np.random.default_rng(d).random(T) + 0.01, T, expect_one=False)
print("checked")
"""
    for scans in ("1", "0"):
        env = dict(os.environ, TGP_MODAL_HEAD_SCANS=scans)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0 and "checked" in r.stdout, (scans, (r.stdout + r.stderr)[-3000:])


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_rand_of_an_lti_model_in_one_launch(tgp, d):
    """rand(rng, model) with the draws supplied (lgssm.jl:65-91) for LTI models up to d = 8: ONE kernel over the draws (k_rand_one: dense
    powers of the open-loop transition, spans with a run-in of `halo` steps) against the oracle's sequential loop, at lengths around every
    tile / span boundary; a slowly mixing model (halo beyond the cap) falls back to the general engine's affine scan."""
    rng = np.random.default_rng(900 + d)
    for T, dt in ((2, 0.1), (77, 0.1), (513, 0.1), (4096, 0.05), (4097, 0.3), (60011, 0.1), (200_003, 0.05)):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), 0.05)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        y_ref = sk.rand(model, *eps)
        dm = device_model(tgp, model)
        y, names = kernels_of(tgp, dm, lambda: tgp.rand(eps, dm))
        assert names == {"k_rand_one"}, (T, dt, names)
        scale = max(1.0, float(np.max(np.abs(y_ref))))
        assert np.max(np.abs(y - y_ref)) <= 1e-9 * scale, (T, dt, np.max(np.abs(y - y_ref)))
    # halo beyond the cap: the general engine
    T = 30_000
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.0005, T), 0.05)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    dm = device_model(tgp, model)
    y, names = kernels_of(tgp, dm, lambda: tgp.rand(eps, dm))
    assert "k_rand_one" not in names, names
    y_ref = sk.rand(model, *eps)
    assert np.max(np.abs(y - y_ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(y_ref))))


def test_rand_of_an_lti_model_with_device_resident_draws(tgp):
    import torch
    T, d = 1_000_003, 3
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    rng = np.random.default_rng(77)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y_ref = sk.rand(model, *eps)
    dm = device_model(tgp, model)
    dev = (torch.as_tensor(eps[0], device="cuda:0"), torch.as_tensor(eps[1], device="cuda:0"), eps[2])
    y, names = kernels_of(tgp, dm, lambda: tgp.rand(dev, dm))
    assert names == {"k_rand_one"}, names
    yy = y.cpu().numpy() if hasattr(y, "cpu") else y
    assert np.max(np.abs(yy - y_ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(y_ref))))


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_filter_of_an_lti_model_in_one_launch(tgp, d):
    """_filter(model, y) (lgssm.jl:171-187) of LTI models up to d = 8: the head on the host, everything behind it in ONE kernel
    (k_filter_one: dense powers of the stationary closed loop) -- against the oracle's literal loop on a short series, against the general
    engine (TGP_OPT_STEADY = 2) on long ones, lengths around every tile / span boundary, and the log marginal likelihood it returns."""
    import ctypes as ct
    from oracle import lgssm_ref as ref
    rng = np.random.default_rng(700 + d)
    T = 1500
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.07)
    y = draw(model, 800 + d)
    fm, fP = ref.filter_(model, y)
    dm = device_model(tgp, model)
    (gm, gP), names = kernels_of(tgp, dm, lambda: tgp._filter(dm, y))
    assert names == {"k_filter_one"}, names
    np.testing.assert_allclose(gm, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(gP, fP, rtol=1e-8, atol=1e-9)
    for T, dt in ((4097, 0.1), (8192, 0.3), (60_011, 0.1), (300_007, 0.05)):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), float(np.exp(rng.uniform(np.log(0.003), np.log(0.5)))))
        y = draw(model, 900 + d)
        dm, dg = device_model(tgp, model), device_model(tgp, model, steady=2)
        (gm, gP), names = kernels_of(tgp, dm, lambda: tgp._filter(dm, y))
        assert names == {"k_filter_one"}, (T, names)
        rm, rP = tgp._filter(dg, y)
        assert np.max(np.abs(gm - rm)) <= 1e-8 * max(1.0, float(np.max(np.abs(rm)))), (T, np.max(np.abs(gm - rm)))
        assert np.max(np.abs(gP - rP)) <= 1e-8 * max(1.0, float(np.max(np.abs(rP)))), (T, np.max(np.abs(gP - rP)))
        # the by-product: logpdf
        hd = dm.handle()
        lml = ct.c_double()
        yy = np.ascontiguousarray(y)
        hd.check(hd.lib.tgp_filter(hd.h, yy.ctypes.data, None, 0, None, None, ct.byref(lml)))
        lp_ref = sk.logpdf(model, y)
        assert abs(lml.value - lp_ref) <= 1e-10 * abs(lp_ref), (T, lml.value, lp_ref)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_posterior_of_an_lti_model_in_one_launch(tgp, d):
    """posterior(model, y) (lgssm.jl:193-221, invert_dynamics :231-238) of Forward LTI models up to d = 8, evaluated: the reverse-time
    transitions of the head on the host, behind it two constant fills and g_(t+1) = m_t - G mu_(t+1) by the filter's ONE kernel -- against the
    oracle's literal loop on a short series, against the general engine (TGP_OPT_STEADY = 2) on long ones (host and device outputs), and
    the smoothing marginals of the evaluated model against the fused call."""
    import torch
    from oracle import lgssm_ref as ref
    rng = np.random.default_rng(1700 + d)
    T = 1500
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.07)
    y = draw(model, 1800 + d)
    post = ref.posterior(model, y)
    dm = device_model(tgp, model)
    dpost, names = kernels_of(tgp, dm, lambda: tgp.posterior(dm, y).materialise())
    assert names == {"k_filter_one"}, names
    for got, want in ((dpost.transitions.As, post["A"]), (dpost.transitions.as_, post["a"]), (dpost.transitions.Qs, post["Q"]),
                      (dpost.x0.m, post["x0m"]), (dpost.x0.P, post["x0P"])):
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-9)
    for T, dt, dev in ((4097, 0.1, False), (8192, 0.3, True), (60_011, 0.1, False), (300_007, 0.05, True)):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), float(np.exp(rng.uniform(np.log(0.003), np.log(0.5)))))
        y = draw(model, 1900 + d)
        dm, dg = device_model(tgp, model), device_model(tgp, model, steady=2)
        yy = torch.as_tensor(y, device="cuda:0") if dev else y
        dpost, names = kernels_of(tgp, dm, lambda: tgp.posterior(dm, yy).materialise())
        assert names == {"k_filter_one"}, (T, names)
        rpost = tgp.posterior(dg, yy).materialise()
        for name in ("As", "as_", "Qs"):
            a, b = getattr(dpost.transitions, name), getattr(rpost.transitions, name)
            a, b = (a.cpu().numpy(), b.cpu().numpy()) if dev else (a, b)
            assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-8 * max(1.0, float(np.max(np.abs(b)))), (T, name, np.max(np.abs(a - b)))
        assert np.max(np.abs(dpost.x0.m - rpost.x0.m)) <= 1e-8 and np.max(np.abs(dpost.x0.P - rpost.x0.P)) <= 1e-8
        # marginals of the evaluated reverse-time model = the fused smoother's (lgssm.jl:111-115 on what :193-221 built)
        if dev:
            continue      # (the model's emission blocks are host arrays: an evaluated model with device transitions does not bind)
        Rn = np.array([0.3])
        mean, var = tgp.posterior_marginals(dm, y, Rn)
        em, ev = tgp.marginals(tgp.replace_observation_noise_cov(dpost, Rn))
        assert np.max(np.abs(em - mean)) <= 1e-7 and np.max(np.abs(ev - var)) <= 1e-7, (T, np.max(np.abs(em - mean)), np.max(np.abs(ev - var)))


def test_a_stream_with_work_of_the_callers_in_front_of_the_kernel(tgp):
    """the one-launch paths talk to the host while their kernel runs (the head of the series is computed on the host: DESIGN 3.15, 3.16); on a
    stream of the caller's with seconds of work queued in front of it the kernel starts late -- the host's side of the hand-over must wait for it
    (it watches the stream, not the clock) and the call must come back with the right numbers"""
    import torch
    T = 20000
    for spec in (KERNELS[3], DECLINED[6]):
        model = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, 3)
        Rn = np.array([0.1])
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
        dm = device_model(tgp, model)
        hd = dm.handle()
        s = torch.cuda.Stream()
        hd.check(hd.lib.tgp_set_stream(hd.h, ctypes.c_void_p(s.cuda_stream)))
        a = torch.randn(6144, 6144, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for _ in range(8):        # a few seconds of fp64 GEMMs in front of the library's kernel (~0.35 s each on an MI355X)
                a = (a @ a) * 1e-4
        lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8
        assert served(dm) > T - 700      # (served by the one-launch path, not a fall-back after a timed-out hand-over)
        torch.cuda.synchronize()
        hd.check(hd.lib.tgp_set_stream(hd.h, None))
        del a


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6])
def test_a_draw_from_the_posterior_in_one_launch(tgp, d):
    """rand(rng, replace_observation_noise_cov(posterior(model, y), Rn)) (posterior_lti_sde.jl:48-58; lgssm.jl:65-91 on the reverse-time model
    of :193-221) of a Forward LTI model WITHOUT evaluating that model: tgp_posterior_rand = k_smooth_one with a noise input, ONE kernel
    (DESIGN 3.17).  Against the oracle's literal loop over the evaluated posterior (T = 5000) and against the product's own evaluated route."""
    from oracle import lgssm_ref as ref
    import torch
    for T in (5000, 9000 + d):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, 10 + d)
        rng = np.random.default_rng(d)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        for Rn in (np.array([0.05]), rng.random(T) * 0.1, None):
            dm = device_model(tgp, model)
            post = tgp.posterior(dm, y)
            if Rn is not None:
                post = tgp.replace_observation_noise_cov(post, Rn)
            got, names = kernels_of(tgp, dm, lambda: tgp.rand(eps, post))
            assert names == {"k_smooth_one<rand>"}, names
            # the evaluated route (tgp_posterior, then tgp_rand on the Reverse model) on a second model object
            dm2 = device_model(tgp, model)
            post2 = tgp.posterior(dm2, y)
            if Rn is not None:
                post2 = tgp.replace_observation_noise_cov(post2, Rn)
            post2.materialise()
            want2 = tgp.rand(eps, post2)
            assert np.max(np.abs(got - want2)) <= 1e-8 * max(1.0, np.max(np.abs(want2)))
            if T == 5000:
                opost = ref.posterior(model, y)
                if Rn is not None:
                    opost = ref.replace_observation_noise_cov(opost, np.broadcast_to(Rn, (T,)).copy())
                want = ref.rand(opost, *eps)
                assert np.max(np.abs(got - want)) <= 1e-8 * max(1.0, np.max(np.abs(want)))
        # device-resident series and draws
        yd = torch.from_numpy(y).cuda()
        ed = (torch.from_numpy(eps[0]).cuda(), torch.from_numpy(eps[1]).cuda(), eps[2])
        dm = device_model(tgp, model)
        gd = tgp.rand(ed, tgp.replace_observation_noise_cov(tgp.posterior(dm, yd), np.array([0.05])))
        dm2 = device_model(tgp, model)
        p2 = tgp.replace_observation_noise_cov(tgp.posterior(dm2, y), np.array([0.05]))
        p2.materialise()
        assert np.max(np.abs(gd.cpu().numpy() - tgp.rand(eps, p2))) <= 1e-8 * max(1.0, float(gd.abs().max()))


def test_a_draw_from_the_posterior_beyond_the_one_launch_path(tgp):
    """d = 8 (the lane's draws no longer fit its registers), a series too short for head + transient: the evaluated route serves the call"""
    from oracle import lgssm_ref as ref
    for spec, T in ((KERNELS[8], 3000), (KERNELS[3], 90)):
        model = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), 0.1)
        d = len(model["x0m"])
        y = draw(model, 4)
        rng = np.random.default_rng(8)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        dm = device_model(tgp, model)
        got = tgp.rand(eps, tgp.posterior(dm, y))
        want = ref.rand(ref.posterior(model, y), *eps)
        assert np.max(np.abs(got - want)) <= 1e-8 * max(1.0, np.max(np.abs(want)))


def test_a_draw_from_the_posterior_with_missing_observations_takes_the_evaluated_route(tgp):
    """NaN == missing in the mirror's convention: the one-launch draw does not cover missing data -- it must step aside (host arrays: the mirror
    sees the NaN; device arrays: the library sees a NaN log marginal likelihood), not return NaNs"""
    from oracle import lgssm_ref as ref
    import torch
    T = 6000
    model = oc.build_lgssm(KERNELS[3], ("regular", 0.0, 0.1, T), 0.1)
    y = draw(model, 4)
    miss = np.zeros(T, dtype=bool)
    miss[[700, 701, 3000]] = True
    yn = np.where(miss, np.nan, y)
    rng = np.random.default_rng(8)
    eps = (rng.standard_normal((T, 3)), rng.standard_normal(T), rng.standard_normal(3))
    want = ref.rand(ref.posterior_missing(model, y, miss), *eps)
    dm = device_model(tgp, model)
    got, names = kernels_of(tgp, dm, lambda: tgp.rand(eps, tgp.posterior(dm, yn)))
    assert "k_smooth_one<rand>" not in names, names
    # (at the missing steps themselves the reference's posterior keeps the 1e15 stand-in variance of missings.jl:55-101 as its emission noise -- its
    #  draws there are ~1e7 eta -- while the mirror's posterior object keeps the prior's noise until replace_observation_noise_cov is called, which
    #  every caller in posterior_lti_sde.jl does: the observed steps are what both define alike)
    assert np.max(np.abs(got - want)[~miss]) <= 1e-8 * max(1.0, np.max(np.abs(want[~miss])))
    # the C entry point itself on a device series with NaNs: unsupported, not NaN
    hd = device_model(tgp, model).handle()
    yd = torch.from_numpy(yn).cuda()
    et, ee = torch.from_numpy(eps[0]).cuda(), torch.from_numpy(eps[1]).cuda()
    out = torch.zeros(T, dtype=torch.float64, device="cuda")
    Rn = torch.tensor([0.1], dtype=torch.float64, device="cuda")
    L = tgp._lib
    torch.cuda.synchronize()
    rc = hd.lib.tgp_posterior_rand(hd.h, L.ptr(yd), L.ptr(Rn), L.ptr(et), L.ptr(ee), L.ptr(np.ascontiguousarray(eps[2])), L.IN_DEVICE | L.OUT_DEVICE | L.SHARED_R, L.ptr(out))
    assert rc == 4, rc      # TGP_EUNSUPPORTED


def test_a_halo_longer_than_a_span(tgp):
    """(the companion of tests/test_smooth_host.py::test_a_halo_longer_than_a_span: the model of the memory fault the randomised sweep found -- modal
    plan: mixes too slowly; dense-powers plan: halo 1520 against spans of 1056, the second workgroup starts behind the head)"""
    spec = ("sum", ("scaled", 0.718, ("stretched", 0.6924527157043311, ("matern12",))), ("scaled", 0.605, ("stretched", 0.6924527157043311, ("matern12",))))
    for T in (60000, 250000):
        model = oc.build_lgssm(spec, ("regular", 0.0, 0.04, T), 2.56e-2)
        y = draw(model, 2)
        Rn = np.exp(np.random.default_rng(0).normal(-2, 1, size=T))
        dm = device_model(tgp, model)
        (lp, mean, var), names = kernels_of(tgp, dm, lambda: tgp.logpdf_and_posterior_marginals(dm, y, Rn))
        assert names == {"k_smooth_one<posterior>"}, names
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8


@pytest.mark.parametrize("d", [1, 3, 4, 6, 8])
def test_a_mean_function_on_a_regular_grid_keeps_the_stationary_gains(tgp, d):
    """every block shared but the emission offset h_t = m(x_t) (lti_sde.jl:118-131): the gains never see the offset, so the model stays on the
    one-launch structure -- k_smooth_one subtracts the offset per step -- instead of the sweep engine (d <= 4) or the general engine"""
    import torch
    for T in (6000, 20003):
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1, ("custom", lambda t: np.sin(0.7 * t) + 0.05 * t))
        assert np.asarray(model["h"]).shape[0] == T
        y = draw(model, 30 + d)
        Rn = np.random.default_rng(d).random(T) * 0.1
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
        dm = device_model(tgp, model)
        lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
        assert names == {"k_smooth_one<logpdf>"}, names
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
        assert names == {"k_smooth_one<posterior>"}, names
        sc = max(1.0, float(np.max(np.abs(m_ref))))
        assert np.max(np.abs(mean - m_ref)) <= 1e-8 * sc and np.max(np.abs(var - v_ref)) <= 1e-8
        yd, rd = torch.from_numpy(y).cuda(), torch.from_numpy(Rn).cuda()
        lp2, mean2, var2 = tgp.logpdf_and_posterior_marginals(dm, yd, rd)
        assert abs(lp2 - lp_ref) <= 1e-10 * abs(lp_ref)
        assert np.max(np.abs(mean2.cpu().numpy() - m_ref)) <= 1e-8 * sc and np.max(np.abs(var2.cpu().numpy() - v_ref)) <= 1e-8
