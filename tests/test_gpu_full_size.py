"""GPU tier at BASELINE.json's FULL sizes. The sequential C oracle (oracle/seq_kalman.c, ~1e7 steps/s on one core) is fast
enough to be run at these sizes directly, so the checks are plain parity, plus two size-independent properties of the
chunked scan: invariance under the chunk size (the scan blocking) and under time sharding.
  cfg2: Matern-5/2 (d = 3), T = 1e7, LTI and per-step layouts   cfg3: sum kernel d = 6 (and d = 5), T = 1e7
  cfg4: d = 4, T = 1e8 (logpdf and posterior marginals; one GPU holds it)
Tolerances: lml relative 1e-10 (it is a sum of 1e7..1e8 terms of mixed sign), marginals absolute 1e-8."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import seq_kalman as sk

pytestmark = pytest.mark.gpu

SPECS = {
    "matern52_d3": ("matern52",),
    "sum52_32_d5": ("sum", ("matern52",), ("matern32",)),
    "sum52_52_d6": ("sum", ("matern52",), ("matern52",)),
    "sum52_12_d4": ("sum", ("matern52",), ("matern12",)),
    "sum52_52_32_d8": ("sum", ("matern52",), ("matern52",), ("matern32",)),     # eight lanes per chunk (tgp_group*.hpp)
    # the same state dimensions with distinct length scales (the one-launch path; two identical summands have no well-conditioned modal form)
    "sum52_52s_d6": ("sum", ("matern52",), ("stretched", 2.0, ("matern52",))),
    "sum52_52s_32_d8": ("sum", ("matern52",), ("stretched", 2.0, ("matern52",)), ("stretched", 0.5, ("matern32",))),
}
# which engine must serve the LTI call of each model (asserted through the kernels' names: a silent fall-back to the general engine
# would still pass the parity checks)
ONE_LAUNCH = {"matern52_d3", "sum52_32_d5", "sum52_12_d4", "sum52_52s_d6", "sum52_52s_32_d8"}


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def _served(model):
    import ctypes
    hd = model.handle()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(a), ctypes.byref(b)))
    return a.value


def _kernels_of(tgp, model, fn):
    hd = model.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    out = fn()
    names = set(hd.profile())
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    return out, names


def _assert_engine(name, per_step, names, model, T):
    if per_step:
        assert any(n.startswith("k_reduce_filter") or n.startswith("k_group_reduce") for n in names) and not any(n.startswith("k_steady") for n in names), names
    elif name in ONE_LAUNCH:
        assert len(names) == 1 and next(iter(names)).startswith(("k_steady_one", "k_lml_stream", "k_post_stream")), names      # (the streaming kernels: logpdf alone, posterior at d <= 3)
        assert _served(model) > T - 700
    else:      # no modal form: ONE kernel on dense powers in both directions (k_smooth_one)
        assert len(names) == 1 and next(iter(names)).startswith("k_smooth_one"), names
        assert _served(model) > T - 700


def _product_model(name, T, per_step=False):
    from temporalgps_jl_amd import lti_sde
    return lti_sde.build_lgssm(lti_sde.to_kernel(SPECS[name]), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1, force_per_step=per_step)


@pytest.mark.parametrize("name,per_step", [("matern52_d3", False), ("matern52_d3", True), ("sum52_32_d5", False), ("sum52_52_d6", False),
                                           ("sum52_52_32_d8", False), ("sum52_52s_d6", False), ("sum52_52s_32_d8", False)])
def test_full_size_parity_with_sequential_oracle(tgp, name, per_step):
    import torch
    T = 10_000_000
    rng = np.random.default_rng(42)
    y = rng.standard_normal(T)
    ref_model = oc.build_lgssm(SPECS[name], ("regular", 0.0, 0.1, T), 0.1)
    lp_ref = sk.logpdf(ref_model, y)
    pm, pv = sk.posterior_marginals(ref_model, y, np.array([1e-18]))
    model = _product_model(name, T, per_step)
    yd = torch.as_tensor(y, device="cuda:0")
    lp, names = _kernels_of(tgp, model, lambda: tgp.logpdf(model, yd))
    _assert_engine(name, per_step, names, model, T)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    (mean, var), names = _kernels_of(tgp, model, lambda: tgp.posterior_marginals(model, yd, Rn))
    _assert_engine(name, per_step, names, model, T)
    assert np.max(np.abs(mean.cpu().numpy() - pm)) <= 1e-8
    assert np.max(np.abs(var.cpu().numpy() - pv)) <= 1e-8
    # chunk-size (scan blocking) invariance at full size (a chunk size set by hand selects the general chunked-scan engine)
    hd = model.handle()
    for chunk in (61, 200):
        hd.set_option(tgp._lib.OPT_CHUNK, chunk)
        assert abs(tgp.logpdf(model, yd) - lp) <= 1e-11 * abs(lp)
        m2, v2 = tgp.posterior_marginals(model, yd, Rn)
        assert float((m2 - mean).abs().max()) <= 1e-9 and float((v2 - var).abs().max()) <= 1e-9


def test_full_size_missing_data_and_sharding_invariance(tgp):
    """cfg2 with 10 % missing observations; the same series through the time-sharded protocol (4 ranks as threads on the
    one GPU, device-resident exchange) must reproduce the single-pass result."""
    import threading

    import torch
    from temporalgps_jl_amd import lti_sde, parallel
    from tests.test_gpu_sharding import _ThreadComm
    from oracle import lgssm_ref as ref
    T, world = 10_000_000, 4
    rng = np.random.default_rng(7)
    y = rng.standard_normal(T)
    miss = rng.random(T) < 0.1
    ref_model = oc.build_lgssm(SPECS["matern52_d3"], ("regular", 0.0, 0.1, T), 0.1)
    R = np.full(T, 0.1)
    R[miss] = 1e15                                                  # missings.jl:25-33
    y0 = np.where(miss, 0.0, y)
    lp_ref = sk.logpdf(dict(ref_model, R=R), y0) + ref.volume_compensation(int(miss.sum()))
    model = _product_model("matern52_d3", T)
    yd, md = torch.as_tensor(y0, device="cuda:0"), torch.as_tensor(miss, device="cuda:0")
    lp, names = _kernels_of(tgp, model, lambda: tgp.logpdf(model, (yd, md)))
    info = model.handle().sweep_info()
    assert info["served"] == 1 and info["attempts"] == 1 and names == {"k_sweep<lti,logpdf>"}, (info, names)    # the sweep engine, one launch
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    # ... and the posterior marginals of the same masked series (missings.jl:25-41 + posterior_lti_sde.jl:27-36), one launch as well
    pm, pv = sk.posterior_marginals(dict(ref_model, R=R), y0, np.array([1e-18]))
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    (mean, var), names = _kernels_of(tgp, model, lambda: tgp.posterior_marginals(model, (yd, md), Rn))
    info = model.handle().sweep_info()
    assert info["served"] == 1 and info["attempts"] == 1 and names == {"k_sweep<lti,posterior>"}, (info, names)
    assert np.max(np.abs(mean.cpu().numpy() - pm)) <= 1e-8
    assert np.max(np.abs(var.cpu().numpy() - pv)) <= 1e-8
    del mean, var, pm, pv
    shared, barrier, out, errs = {}, threading.Barrier(world), {}, []
    lib = tgp._lib.load()
    for ph in (0, 1):
        n = lib.tgp_shard_slot_size(ph, 3)
        shared[("g", n)] = torch.zeros(world * n, dtype=torch.float64, device="cuda:0")
    shared[("r", 4)] = torch.zeros(world, 4, dtype=torch.float64, device="cuda:0")

    def run(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = parallel.segment_bounds(T, world, rank)
            seg = lti_sde.build_lgssm(lti_sde.to_kernel(SPECS["matern52_d3"]), lti_sde.RegularSpacing(0.1 * lo, 0.1, hi - lo), 0.1)
            sh = parallel.ShardedLGSSM(seg, world, rank, engine=parallel.HIPEngine(seg), comm=_ThreadComm(world, rank, shared, barrier))
            out[rank] = sh.logpdf((yd[lo:hi], md[lo:hi]))
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
            barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for r in range(world):
        assert abs(out[r] - lp) <= 1e-11 * abs(lp)


def test_cfg4_T1e8_d4_logpdf_and_posterior_marginals(tgp):
    """BASELINE config 4's series on ONE GPU (it is time-sharded there): T = 1e8, d = 4."""
    import torch
    T = 100_000_000
    rng = np.random.default_rng(3)
    y = rng.standard_normal(T)
    ref_model = oc.build_lgssm(SPECS["sum52_12_d4"], ("regular", 0.0, 0.1, T), 0.1)
    lp_ref = sk.logpdf(ref_model, y)
    model = _product_model("sum52_12_d4", T)
    yd = torch.as_tensor(y, device="cuda:0")
    lp, names = _kernels_of(tgp, model, lambda: tgp.logpdf(model, yd))
    _assert_engine("sum52_12_d4", False, names, model, T)
    assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
    # ... and the posterior marginals of the same 1e8-step series (one combined call), against the sequential oracle
    pm, pv = sk.posterior_marginals(ref_model, y, np.array([1e-18]))
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    lp2, mean, var = tgp.logpdf_and_posterior_marginals(model, yd, Rn)
    assert lp2 == lp or abs(lp2 - lp_ref) <= 1e-10 * abs(lp_ref)
    assert float(torch.max(torch.abs(mean - torch.as_tensor(pm, device="cuda:0")))) <= 1e-8
    assert float(torch.max(torch.abs(var - torch.as_tensor(pv, device="cuda:0")))) <= 1e-8


def test_full_size_filter_and_evaluated_posterior(tgp):
    """cfg2's series through the rest of the LTI interface (one launch each): `_filter` (lgssm.jl:171-187) and the evaluated
    `posterior` (lgssm.jl:193-221).  Size-independent ties instead of a second oracle: behind the head the reverse-time transition is ONE
    matrix and its offsets are g_(t+1) = (I - G A) m_t - G a in the filter's own means; the marginals of the evaluated model
    (lgssm.jl:111-115, the general engine on T x (2 d^2 + d) doubles) are the smoother's -- checked against the sequential C oracle; the
    last smoothed marginal is the last filtered one."""
    T, name = 10_000_000, "matern52_d3"
    rng = np.random.default_rng(4242)
    y = rng.standard_normal(T)
    ref_model = oc.build_lgssm(SPECS[name], ("regular", 0.0, 0.1, T), 0.1)
    Rn = np.array([0.3])
    pm, pv = sk.posterior_marginals(ref_model, y, Rn)
    model = _product_model(name, T)
    (m, P), names = _kernels_of(tgp, model, lambda: tgp._filter(model, y))
    assert names == {"k_filter_one"}, names
    dpost, names = _kernels_of(tgp, model, lambda: tgp.posterior(model, y).materialise())
    assert names == {"k_filter_one"}, names
    G, g, L = dpost.transitions.As, dpost.transitions.as_, dpost.transitions.Qs
    assert G.shape == (T, 3, 3) and g.shape == (T, 3) and L.shape == (T, 3, 3)
    assert np.array_equal(G[1000:], np.broadcast_to(G[-1], G[1000:].shape)) and np.array_equal(L[1000:], np.broadcast_to(L[-1], L[1000:].shape))
    A, a = np.asarray(ref_model["A"]).reshape(-1, 3, 3)[0], np.asarray(ref_model["a"]).reshape(-1, 3)[0]
    M = np.eye(3) - G[-1] @ A
    tie = m[999:-1] @ M.T - G[-1] @ a
    assert np.max(np.abs(g[1000:] - tie)) <= 1e-9 * max(1.0, float(np.max(np.abs(g)))), np.max(np.abs(g[1000:] - tie))
    np.testing.assert_allclose(dpost.x0.m, m[-1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(dpost.x0.P, P[-1], rtol=0, atol=1e-12)
    mean, var = tgp.marginals(tgp.replace_observation_noise_cov(dpost, Rn))
    assert np.max(np.abs(mean - pm)) <= 1e-8 and np.max(np.abs(var - pv)) <= 1e-8, (np.max(np.abs(mean - pm)), np.max(np.abs(var - pv)))
    H, h = np.asarray(ref_model["H"]).reshape(-1, 3)[0], float(np.asarray(ref_model["h"]).reshape(-1)[0])
    assert abs(H @ m[-1] + h - pm[-1]) <= 1e-9 and abs(H @ P[-1] @ H + Rn[0] - pv[-1]) <= 1e-9


def test_full_size_rand_against_the_sequential_oracle(tgp):
    """rand with the draws supplied (lgssm.jl:65-91) at cfg2's size: the one-launch kernel on 1e7 x (d + 1) draws against the oracle's loop."""
    T, name, d = 10_000_000, "matern52_d3", 3
    rng = np.random.default_rng(777)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    ref_model = oc.build_lgssm(SPECS[name], ("regular", 0.0, 0.1, T), 0.1)
    y_ref = sk.rand(ref_model, *eps)
    model = _product_model(name, T)
    y, names = _kernels_of(tgp, model, lambda: tgp.rand(eps, model))
    assert names == {"k_rand_one"}, names
    assert np.max(np.abs(y - y_ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(y_ref))))
