"""GPU tier: the in-library multi-GPU handle (tgp_create_multi ..., csrc/tgp_multi.hip) against the unsharded oracle.

The box has ONE GPU: ranks that share cuda:0 exercise the whole protocol (worker threads, phase boundaries, the
event-ordered copy transport, folds on the device); a 1-rank handle goes through RCCL proper (ncclCommInitAll,
ncclAllGather on the handle's stream); distinct devices are used when the box has them."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk
from tests import _util as U

pytestmark = pytest.mark.gpu

LOGPDF_RTOL, MARGINAL_ATOL = 1e-10, 1e-8


def _dev_model(tgp, model):
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    return tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])


def _check(ms, model, y, Rnew, mask=None):
    yy = y if mask is None else (y, mask)
    if mask is None:
        lp_ref = sk.logpdf(model, y)
        pm, pv = sk.posterior_marginals(model, y, Rnew)
    else:
        lp_ref = ref.logpdf_missing(model, y, mask)
        post = ref.posterior_missing(model, y, mask)
        pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, np.broadcast_to(Rnew, (model["T"],))))
    lp = ms.logpdf(yy)
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref), (lp, lp_ref)
    mean, var = ms.posterior_marginals(yy, Rnew)
    assert np.max(np.abs(mean - pm)) <= MARGINAL_ATOL and np.max(np.abs(var - pv)) <= MARGINAL_ATOL
    lp2, mean2, var2 = ms.logpdf_and_posterior_marginals(yy, Rnew)
    assert abs(lp2 - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)
    assert np.array_equal(mean2, mean) and np.array_equal(var2, var)
    assert ms.logpdf(yy) == lp          # a second round on the same handles (carry-ins replaced again)


@pytest.mark.parametrize("ndev", [1, 2, 3, 5])
def test_lti_series_over_ranks_sharing_one_gpu(ndev):
    import temporalgps_jl_amd as tgp
    T = 120_007
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = np.random.default_rng(11).standard_normal(T)
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    assert ms.transport.startswith("rccl" if ndev == 1 else "copy"), ms.transport
    _check(ms, model, y, np.array([0.05]))


def test_rccl_transport_single_rank_goes_through_ncclAllGather():
    """ndev = 1 on distinct devices [0]: ncclCommInitAll + ncclAllGather on the handle's stream (the transport string says which)."""
    import temporalgps_jl_amd as tgp
    T = 50_000
    model = oc.build_lgssm(("sum", ("matern52",), ("matern12",)), ("regular", 0.0, 0.1, T), 0.1)      # d = 4 (BASELINE config 4's model)
    y = np.random.default_rng(12).standard_normal(T)
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0])
    assert ms.transport == "rccl", ms.transport
    _check(ms, model, y, np.array([1e-18]))


def test_distinct_devices_when_the_box_has_them():
    import torch
    import temporalgps_jl_amd as tgp
    n = min(2, torch.cuda.device_count())
    if n < 2:
        pytest.skip("one GPU visible: the distinct-device RCCL group needs two")
    T = 200_001
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = np.random.default_rng(13).standard_normal(T)
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=list(range(n)))
    assert ms.transport == "rccl", ms.transport
    _check(ms, model, y, np.array([0.05]))


@pytest.mark.parametrize("d", [1, 2, 5])
def test_per_step_model_with_missing_observations(d):
    """explicit per-step blocks (sliced per segment by tgp_multi_model_set), per-step R_new, a missing mask"""
    import temporalgps_jl_amd as tgp
    T, ndev = 4_001, 3
    rng = np.random.default_rng(20 + d)
    model = U.random_lgssm(rng, True, d, T)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    mask = rng.random(T) < 0.1
    Rnew = rng.random(T) + 0.05
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    lp_ref = ref.logpdf_missing(model, y, mask)
    lp = ms.logpdf((y, mask))
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)
    post = ref.posterior_missing(model, y, mask)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, Rnew))
    mean, var = ms.posterior_marginals((y, mask), Rnew)
    assert np.max(np.abs(mean - pm)) <= MARGINAL_ATOL and np.max(np.abs(var - pv)) <= MARGINAL_ATOL


def test_device_resident_segments():
    """one CUDA tensor per rank in, one per rank out (TGP_IN_DEVICE | TGP_OUT_DEVICE)"""
    import torch
    import temporalgps_jl_amd as tgp
    T, ndev = 90_001, 3
    model = oc.build_lgssm(("matern32",), ("regular", 0.0, 0.1, T), 0.1)
    y = np.random.default_rng(14).standard_normal(T)
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    parts = [torch.as_tensor(y[lo:hi], device="cuda:0") for lo, hi in ms.bounds]
    lp_ref = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, np.array([0.05]))
    lp, mean, var = ms.logpdf_and_posterior_marginals(parts, np.array([0.05]))
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)
    mean, var = (np.concatenate([t.cpu().numpy() for t in x]) for x in (mean, var))
    assert np.max(np.abs(mean - pm)) <= MARGINAL_ATOL and np.max(np.abs(var - pv)) <= MARGINAL_ATOL


@pytest.mark.timeout(120)
def test_a_failing_rank_stops_every_rank():
    """a segment whose innovation variance is not positive: TGP_ENOTPD from the call, no rank left waiting at a phase boundary,
    and the handle serves the next (valid) call"""
    import temporalgps_jl_amd as tgp
    T, ndev, d = 3_000, 3, 2
    rng = np.random.default_rng(15)
    model = U.random_lgssm(rng, True, d, T)
    y = rng.standard_normal(T)
    bad = dict(model, R=model["R"].copy())
    bad["R"][T // 2] = -1e6           # inside rank 1's segment
    ms = tgp.MultiLGSSM(_dev_model(tgp, bad), devices=[0] * ndev)
    with pytest.raises(tgp._lib.NotPositiveDefinite):
        ms.logpdf(y)
    with pytest.raises(tgp._lib.NotPositiveDefinite):
        ms.posterior_marginals(y, np.array([0.1]))
    good = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    lp_ref = ref.logpdf(model, y)
    assert abs(good.logpdf(y) - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)


def test_unsupported_models_are_refused():
    import temporalgps_jl_amd as tgp
    rng = np.random.default_rng(16)
    rev = U.random_lgssm(rng, False, 2, 100, "R")
    tr = tgp.GaussMarkovModel(tgp.Reverse, rev["A"], rev["a"], rev["Q"], tgp.Gaussian(rev["x0m"], rev["x0P"]))
    with pytest.raises(tgp._lib.Unsupported):
        tgp.MultiLGSSM(tgp.LGSSM(tr, tgp.ScalarOutputLGC(rev["H"], rev["h"], rev["R"]), T=100), devices=[0, 0])
    m = U.random_lgssm(rng, False, 2, 3)
    with pytest.raises(tgp._lib.TGPError):          # fewer steps than ranks
        tgp.MultiLGSSM(_dev_model(tgp, m), devices=[0] * 4)


def test_bench_launched_directly_with_several_gpus_uses_the_in_library_handle():
    """`python bench.py --gpus 2` without a torch.distributed environment: ONE process, tgp_create_multi (here both ranks on cuda:0)."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(U.ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--steps", "2", "--warmup", "1", "--T", "400000"]
    res = subprocess.run(cmd, env=env, cwd=U.ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["T"] == 400000
    assert out["config"]["backend"].startswith("copy") and "tgp_create_multi" in out["config"]["parallelism"]
    assert out["value"] > 0 and np.isfinite(out["config"]["logpdf"])
    assert out["single_gpu_reference"]["value"] > 0


def _kernels_of_rank(ms, tgp, rank, fn):
    ms.mh.set_option(tgp._lib.OPT_PROFILE, 1)
    for r in range(ms.W):
        ms.mh.lib.tgp_profile_reset(ms.mh.lib.tgp_multi_handle(ms.mh.m, r))
    out = fn()
    ms.mh.set_option(tgp._lib.OPT_PROFILE, 0)
    return out, set(ms.mh.rank_profile(rank))


@pytest.mark.parametrize("steady_opt", [2, 3])
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8])
def test_lti_shards_run_on_the_stationary_gain_engine(d, steady_opt):
    """An LTI model's shards take the stationary-gain engine's two-half calls (ONE all-gather; head on rank 0 only, tail on the last
    rank only, segments aligned to 512-step tiles): same results as the oracle, and the kernels that ran say which engine it was."""
    import temporalgps_jl_amd as tgp
    rng = np.random.default_rng(60 + d)
    T, ndev = 70_001, 3
    model = U.random_lgssm(rng, False, d, T)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    ms.mh.set_option(tgp._lib.OPT_STEADY, steady_opt)      # 3 (default): the one-launch path's segments first; 2: the five-launch engine's shards
    assert all(lo % 512 == 0 for lo, _ in ms.bounds)
    lp_ref = sk.logpdf(model, y)
    Rnew = rng.random(T) + 0.05
    pm, pv = sk.posterior_marginals(model, y, Rnew)
    one = None
    for rank in (0, 1, 2):
        lp, names = _kernels_of_rank(ms, tgp, rank, lambda: ms.logpdf(y))
        assert not any(n.startswith("k_reduce_filter") for n in names), names
        if one is None:
            one = all(n.startswith("k_steady_one") for n in names)
        if one:      # ONE kernel per rank, nothing of the shard protocol (the plan's verdict is the same for every rank)
            assert steady_opt == 3 and names and all(n.startswith("k_steady_one") for n in names), names
        else:
            assert "k_steady_shard_fold" in names, names
        assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)
    (lp, mean, var), names = _kernels_of_rank(ms, tgp, 1, lambda: ms.logpdf_and_posterior_marginals(y, Rnew))
    if one:
        assert all(n.startswith("k_steady_one") and "posterior" in n for n in names), names
    else:
        assert "k_steady_apply<posterior>" in names and "k_steady_shard_pack" in names, names
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)
    assert np.max(np.abs(mean - pm)) <= MARGINAL_ATOL and np.max(np.abs(var - pv)) <= MARGINAL_ATOL
    mean1, var1 = ms.posterior_marginals(y, np.array([0.3]))
    pm1, pv1 = sk.posterior_marginals(model, y, np.array([0.3]))
    assert np.max(np.abs(mean1 - pm1)) <= MARGINAL_ATOL and np.max(np.abs(var1 - pv1)) <= MARGINAL_ATOL


def test_segments_the_stationary_gain_engine_cannot_take_fall_back_to_the_general_protocol():
    """short segments (interior boundaries not on tile multiples) and a series with missing observations: every rank agrees through the
    gathered elements that the engine does not apply, the general protocol serves the call, later calls go there directly"""
    import temporalgps_jl_amd as tgp
    T, ndev = 9_000, 3                 # 3000-step segments: not aligned (alignment starts at 4096 steps per rank)
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    y = np.random.default_rng(17).standard_normal(T)
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    assert any(lo % 512 for lo, _ in ms.bounds)
    lp, names = _kernels_of_rank(ms, tgp, 1, lambda: ms.logpdf(y))
    assert any(n.startswith("k_reduce_filter") for n in names), names          # the general engine served it ...
    lp2, names2 = _kernels_of_rank(ms, tgp, 1, lambda: ms.logpdf(y))
    assert not any(n.startswith("k_steady") for n in names2), names2           # ... and the second call does not try again
    lp_ref = sk.logpdf(model, y)
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref) and lp2 == lp
    _check(ms, model, y, np.array([0.05]))
    # missing data on aligned segments: the general protocol at once
    T = 60_000
    model = oc.build_lgssm(("matern32",), ("regular", 0.0, 0.1, T), 0.1)
    y = np.random.default_rng(18).standard_normal(T)
    mask = np.random.default_rng(19).random(T) < 0.05
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    lp, names = _kernels_of_rank(ms, tgp, 0, lambda: ms.logpdf((y, mask)))
    assert not any(n.startswith("k_steady") for n in names)
    lp_ref = ref.logpdf_missing(model, y, mask)
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)


@pytest.mark.parametrize("dt,ndev", [(0.004, 3), (0.01, 2)])
def test_slowly_mixing_lti_shards_take_the_scanned_carries(dt, ndev):
    """A filter that forgets slowly (dt = 0.004 at unit length scale: Phi^4096 has NOT decayed, the covariance needs ~1000 steps to
    settle, the head spans several tiles): the workgroup carries come from k_carry's scans, a shard's lam enters them at the ragged end
    (G^(last workgroup's steps), the lane that owns the last element). Against the oracle, and the engine must really have served it."""
    import temporalgps_jl_amd as tgp
    T = 150_011
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, dt, T), 0.1)
    rng = np.random.default_rng(71)
    y = sk.rand(model, rng.standard_normal((T, 3)), rng.standard_normal(T), rng.standard_normal(3))
    ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * ndev)
    ms.mh.set_option(tgp._lib.OPT_STEADY, 2)          # (the five-launch engine's shards are what is under test; the one-launch path serves dt = 0.01 too)
    (lp, mean, var), names = _kernels_of_rank(ms, tgp, ndev - 1, lambda: ms.logpdf_and_posterior_marginals(y, np.array([0.02])))
    assert "k_steady_shard_fold" in names and not any(n.startswith("k_reduce_filter") for n in names), names
    lp_ref = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, np.array([0.02]))
    assert abs(lp - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)
    assert np.max(np.abs(mean - pm)) <= MARGINAL_ATOL and np.max(np.abs(var - pv)) <= MARGINAL_ATOL
    assert abs(ms.logpdf(y) - lp_ref) <= LOGPDF_RTOL * abs(lp_ref)


@pytest.mark.parametrize("ndev,steady_opt", [(2, 2), (3, 2), (4, 3)])
def test_rccl_branch_with_several_ranks_through_a_stub_library(ndev, steady_opt, tmp_path):
    """The RCCL branch of the handle with W > 1 -- one worker thread per rank calling ncclAllGather on its own communicator and stream, no
    group call -- on a box with one GPU: tests/stub_rccl.cpp stands in for librccl (TGP_MULTI_RCCL_LIB, TGP_MULTI_TRANSPORT=rccl; a
    subprocess, so that the environment reaches the library's first look at it).  What it checks: the handle's side of the protocol --
    slots, counts, stream ordering, the fold of the gathered elements -- not RCCL itself."""
    import os
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc to build the stub library")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = str(tmp_path / "libstub_rccl.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-o", stub, os.path.join(root, "tests", "stub_rccl.cpp")], check=True, capture_output=True)
    code = f"""
import ctypes, sys
import numpy as np
sys.path.insert(0, {root!r})
import temporalgps_jl_amd as tgp
from oracle import components as oc
from oracle import seq_kalman as sk
from tests.test_gpu_multi import _dev_model, _check
T = 90_011
model = oc.build_lgssm(("sum", ("matern52",), ("matern12",)), ("regular", 0.0, 0.1, T), 0.1)
y = np.random.default_rng(21).standard_normal(T)
ms = tgp.MultiLGSSM(_dev_model(tgp, model), devices=[0] * {ndev})
ms.mh.set_option(tgp._lib.OPT_STEADY, {steady_opt})
assert ms.transport == "rccl", ms.transport
_check(ms, model, y, np.array([0.05]))
# a per-step model with missing observations: the general protocol (two all-gathers per call)
from tests import _util as U
T2 = 4_001
m2 = U.random_lgssm(np.random.default_rng(22), True, 2, T2)
y2 = np.random.default_rng(23).standard_normal(T2)
mask = np.random.default_rng(24).random(T2) < 0.1
ms2 = tgp.MultiLGSSM(_dev_model(tgp, m2), devices=[0] * {ndev})
assert ms2.transport == "rccl", ms2.transport
_check(ms2, m2, y2, np.array([0.05]), mask=mask)
calls = ctypes.CDLL({stub!r}).stub_rccl_total_calls()
print("ncclAllGather calls:", calls)
assert calls >= 2 * {ndev}, calls
"""
    env = dict(os.environ, TGP_MULTI_RCCL_LIB=stub, TGP_MULTI_TRANSPORT="rccl")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ncclAllGather calls:" in r.stdout
