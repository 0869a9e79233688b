"""GPU tier: the time-sharded path with the REAL HIP engine (tgp_segment_reduce / tgp_smoother_forward /
tgp_smoother_backward / TGP_REUSE_REDUCE) -- W ranks as W processes sharing cuda:0, exchanging through gloo
(the single-GPU box cannot host an RCCL group of W > 1; the collective payload is a few hundred bytes of
host data either way). Compared with the unsharded oracle."""
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import components as oc
from oracle import seq_kalman as sk

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, T, layout, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        import temporalgps_jl_amd as tgp
        from temporalgps_jl_amd import lti_sde, parallel
        rng = np.random.default_rng(5)
        y_all = rng.standard_normal(T)
        lo, hi = parallel.segment_bounds(T, world, rank)
        model = lti_sde.build_lgssm(lti_sde.Matern52Kernel(), lti_sde.RegularSpacing(0.0, 0.1, hi - lo), 0.1,
                                    force_per_step=(layout == "per_step"))
        sh = parallel.ShardedLGSSM(model, world, rank)
        y = torch.as_tensor(y_all[lo:hi], device="cuda:0")
        lp = sh.logpdf(y)
        mean, var = sh.posterior_marginals(y, np.array([0.05]))
        lp2 = sh.logpdf(y)                      # a second round on the same handle (carry-in replaced again)
        ret[rank] = (lp, lp2, lo, hi, mean.cpu().numpy(), var.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layout", [(2, "lti"), (3, "lti"), (2, "per_step")])
def test_sharded_hip_engine_equals_sequential(world, layout):
    T = 200_003
    rng = np.random.default_rng(5)
    y = rng.standard_normal(T)
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    lp_ref = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, np.array([0.05]))
    ret = mp.Manager().dict()
    port = 29700 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, T, layout, ret), nprocs=world, join=True)
    mean, var = np.zeros(T), np.zeros(T)
    for r in range(world):
        lp, lp2, lo, hi, m, v = ret[r]
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        assert lp2 == lp
        mean[lo:hi], var[lo:hi] = m, v
    assert np.max(np.abs(mean - pm)) <= 1e-8
    assert np.max(np.abs(var - pv)) <= 1e-8


def test_sharded_path_over_rccl_single_rank():
    """The sharded code path with the REAL collectives backend (nccl == RCCL) on the one GPU this box has: a 1-rank group
    still goes through segment reduce -> all_gather_into_tensor -> fold -> local passes -> all_reduce."""
    import torch
    import temporalgps_jl_amd as tgp  # noqa: F401
    from temporalgps_jl_amd import lti_sde, parallel
    port = 29900 + (os.getpid() % 1000)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        T = 100_000
        rng = np.random.default_rng(6)
        y_np = rng.standard_normal(T)
        model = lti_sde.build_lgssm(lti_sde.Matern52Kernel(), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
        sh = parallel.ShardedLGSSM(model, 1, 0, engine=parallel.HIPEngine(model))
        y = torch.as_tensor(y_np, device="cuda:0")
        lp = sh.logpdf(y)
        mean, var = sh.posterior_marginals(y, np.array([0.05]))
        ref_model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
        lp_ref = sk.logpdf(ref_model, y_np)
        pm, pv = sk.posterior_marginals(ref_model, y_np, np.array([0.05]))
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        assert np.max(np.abs(mean.cpu().numpy() - pm)) <= 1e-8 and np.max(np.abs(var.cpu().numpy() - pv)) <= 1e-8
    finally:
        dist.destroy_process_group()


class _ThreadComm:
    """In-process stand-in for RCCL so the DEVICE-RESIDENT exchange (tgp_shard_* + k_fold) can be exercised with W > 1
    on the one GPU this box has: W ranks are W threads sharing cuda:0, each with its own HIP stream; a collective is a
    device-to-device copy into a shared tensor between two barriers (the stream is drained before each barrier, which
    RCCL would not need)."""

    def __init__(self, world, rank, shared, barrier):
        self.world, self.rank, self.shared, self.barrier = world, rank, shared, barrier

    def all_gather(self, gathered, slot):
        import torch
        n = slot.numel()
        buf = self.shared.setdefault(("g", n), torch.zeros(self.world * n, dtype=torch.float64, device=slot.device))
        buf[self.rank * n:(self.rank + 1) * n].copy_(slot)
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()
        gathered.copy_(buf)
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()

    def all_reduce_sum(self, t):
        import torch
        n = t.numel()
        buf = self.shared.setdefault(("r", n), torch.zeros(self.world, n, dtype=torch.float64, device=t.device))
        buf[self.rank].copy_(t)
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()
        t.copy_(buf.sum(0))
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()


@pytest.mark.parametrize("world,layout,d_kernel", [(2, "lti", "matern52"), (4, "lti", "matern32"), (3, "per_step", "matern52"),
                                                   (3, "lti", "sum52_52")])
def test_device_resident_exchange_threads(world, layout, d_kernel):
    import threading

    import torch
    import temporalgps_jl_amd as tgp  # noqa: F401
    from temporalgps_jl_amd import lti_sde, parallel
    T = 150_001
    rng = np.random.default_rng(8)
    y_all = rng.standard_normal(T)
    y_all[rng.random(T) < 0.05] = np.nan
    kern = {"matern52": lti_sde.Matern52Kernel(), "matern32": lti_sde.Matern32Kernel(),
            "sum52_52": lti_sde.Matern52Kernel() + 0.5 * lti_sde.Matern52Kernel().stretch(0.3)}[d_kernel]
    spec = {"matern52": ("matern52",), "matern32": ("matern32",),
            "sum52_52": ("sum", ("matern52",), ("scaled", 0.5, ("stretched", 0.3, ("matern52",))))}[d_kernel]
    ref_model = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), 0.1)
    from oracle import lgssm_ref as ref
    m2, y2, nmiss = ref.transform_model_and_obs(ref_model, y_all, np.isnan(y_all))      # missings.jl:25-33
    lp_ref = sk.logpdf(m2, y2) + ref.volume_compensation(nmiss)
    pm, pv = sk.posterior_marginals(m2, y2, np.array([0.05]))
    shared, barrier, out, errs = {}, threading.Barrier(world), {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = parallel.segment_bounds(T, world, rank)
            model = lti_sde.build_lgssm(kern, lti_sde.RegularSpacing(0.1 * lo, 0.1, hi - lo), 0.1, force_per_step=(layout == "per_step"))
            sh = parallel.ShardedLGSSM(model, world, rank, engine=parallel.HIPEngine(model),
                                       comm=_ThreadComm(world, rank, shared, barrier))
            y = torch.as_tensor(np.nan_to_num(y_all[lo:hi]), device="cuda:0")
            mask = torch.as_tensor(np.isnan(y_all[lo:hi]), device="cuda:0")
            lp = sh.logpdf((y, mask))
            mean, var = sh.posterior_marginals((y, mask), np.array([0.05]))
            lp2 = sh.logpdf((y, mask))
            out[rank] = (lp, lp2, lo, hi, mean.cpu().numpy(), var.cpu().numpy())
        except Exception as ex:          # noqa: BLE001
            errs.append(ex)
            barrier.abort()

    # the shared exchange tensors are created up front (dict.setdefault from several threads would race)
    d = ref_model["A"].shape[-1]
    from temporalgps_jl_amd import _lib
    lib = _lib.load()
    for ph in (0, 1):
        n = lib.tgp_shard_slot_size(ph, d)
        shared[("g", n)] = torch.zeros(world * n, dtype=torch.float64, device="cuda:0")
    shared[("r", 4)] = torch.zeros(world, 4, dtype=torch.float64, device="cuda:0")
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    mean, var = np.zeros(T), np.zeros(T)
    for r in range(world):
        lp, lp2, lo, hi, m, v = out[r]
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)
        assert lp2 == lp
        mean[lo:hi], var[lo:hi] = m, v
    assert np.max(np.abs(mean - pm)) <= 1e-8
    assert np.max(np.abs(var - pv)) <= 1e-8


@pytest.mark.parametrize("world,d_kernel,steady_opt", [(2, "matern52", 3), (4, "matern32", 3), (3, "sum52_52", 3), (3, "matern52", 2), (2, "sum52_52", 2)])
def test_device_resident_exchange_threads_on_the_stationary_gain_engine(world, d_kernel, steady_opt):
    """The one-process-per-GPU driver (parallel.ShardedLGSSM) with an LTI series and no missing data: the shards run the stationary-gain
    engine's two-half calls (tgp_shard_steady_begin / _finish, ONE all-gather) -- checked through the kernels' names -- and a second
    case whose segments are too short for it agrees, through the gathered elements, to take the general protocol."""
    import threading

    import torch
    import temporalgps_jl_amd as tgp
    from temporalgps_jl_amd import _lib, lti_sde, parallel
    kern = {"matern52": lti_sde.Matern52Kernel(), "matern32": lti_sde.Matern32Kernel(),
            "sum52_52": lti_sde.Matern52Kernel() + 0.5 * lti_sde.Matern52Kernel().stretch(0.3)}[d_kernel]
    spec = {"matern52": ("matern52",), "matern32": ("matern32",),
            "sum52_52": ("sum", ("matern52",), ("scaled", 0.5, ("stretched", 0.3, ("matern52",))))}[d_kernel]
    lib = _lib.load()
    for T, expect_steady in ((150_001, True), (6_000, False)):
        rng = np.random.default_rng(9)
        y_all = rng.standard_normal(T)
        ref_model = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), 0.1)
        d = ref_model["A"].shape[-1]
        lp_ref = sk.logpdf(ref_model, y_all)
        pm, pv = sk.posterior_marginals(ref_model, y_all, np.array([0.05]))
        shared, barrier, out, errs = {}, threading.Barrier(world), {}, []

        def run(rank):
            try:
                torch.cuda.set_device(0)
                lo, hi = parallel.segment_bounds(T, world, rank)
                model = lti_sde.build_lgssm(kern, lti_sde.RegularSpacing(0.1 * lo, 0.1, hi - lo), 0.1)
                model.handle_options[tgp._lib.OPT_STEADY] = steady_opt      # 3: the one-launch path's segments; 2: the five-launch engine's shards
                sh = parallel.ShardedLGSSM(model, world, rank, engine=parallel.HIPEngine(model), comm=_ThreadComm(world, rank, shared, barrier))
                hd = model.handle()
                hd.set_option(tgp._lib.OPT_PROFILE, 1)
                y = torch.as_tensor(y_all[lo:hi], device="cuda:0")
                lp = sh.logpdf(y)
                lp3, mean, var = sh.logpdf_and_posterior_marginals(y, np.array([0.05]))
                out[rank] = (lp, lp3, lo, hi, mean.cpu().numpy(), var.cpu().numpy(), set(hd.profile()))
            except Exception as ex:          # noqa: BLE001
                errs.append(ex)
                barrier.abort()
        for n in (lib.tgp_shard_slot_size(0, d), lib.tgp_shard_slot_size(1, d), lib.tgp_shard_steady_slot_size(d), 1):
            shared[("g", n)] = torch.zeros(world * n, dtype=torch.float64, device="cuda:0")
        for hh in range(16, 1537, 16):          # (the edge exchange of the one-launch segments: 2 halo values per rank)
            shared[("g", 2 * hh)] = torch.zeros(world * 2 * hh, dtype=torch.float64, device="cuda:0")
        shared[("r", 4)] = torch.zeros(world, 4, dtype=torch.float64, device="cuda:0")
        shared[("r", 1)] = torch.zeros(world, 1, dtype=torch.float64, device="cuda:0")
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        mean, var = np.zeros(T), np.zeros(T)
        for r in range(world):
            lp, lp3, lo, hi, m, v, names = out[r]
            assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref) and abs(lp3 - lp_ref) <= 1e-10 * abs(lp_ref)
            general = any(n.startswith("k_reduce_filter") for n in names)
            if steady_opt == 3 and any(n.startswith("k_steady_one") for n in names):      # ONE kernel per rank and call, nothing of the shard protocol
                assert all(n.startswith("k_steady_one") for n in names), (T, r, names)           # (its segments may be shorter than the shards')
            elif expect_steady:
                assert "k_steady_shard_fold" in names and not general, (T, r, names)
            else:            # (the first call tried the engine -- its kernels are in the profile -- and every rank fell back together)
                assert general, (T, r, names)
            mean[lo:hi], var[lo:hi] = m, v
        assert np.max(np.abs(mean - pm)) <= 1e-8 and np.max(np.abs(var - pv)) <= 1e-8


def test_nan_in_one_segment_only_sends_every_rank_to_the_general_protocol():
    """A host series whose NaNs (== missing) fall in ONE rank's segment: that rank cannot take the stationary-gain shard call, the others could
    -- all of them must agree (through the gathered elements) on the general protocol, or the collectives that follow differ in size."""
    import threading

    import torch
    import temporalgps_jl_amd as tgp
    from temporalgps_jl_amd import _lib, lti_sde, parallel
    lib = _lib.load()
    from oracle import lgssm_ref as ref
    world, T = 3, 12_000                                       # (the pure-Python oracle handles the missing mask: a short series)
    rng = np.random.default_rng(4)
    y_all = rng.standard_normal(T)
    miss = np.zeros(T, dtype=bool)
    miss[T // 2 + 17] = miss[T // 2 + 400] = True           # the middle rank's segment only
    y_nan = y_all.copy()
    y_nan[miss] = np.nan
    ref_model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, T), 0.1)
    d = 3
    lp_ref = ref.logpdf_missing(ref_model, y_all, miss)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior_missing(ref_model, y_all, miss), np.array([0.05])))
    shared, barrier, out, errs = {}, threading.Barrier(world), {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = parallel.segment_bounds(T, world, rank)
            model = lti_sde.build_lgssm(lti_sde.Matern52Kernel(), lti_sde.RegularSpacing(0.1 * lo, 0.1, hi - lo), 0.1)
            sh = parallel.ShardedLGSSM(model, world, rank, engine=parallel.HIPEngine(model), comm=_ThreadComm(world, rank, shared, barrier))
            lp = sh.logpdf(y_nan[lo:hi])                      # host slices: NaN == missing
            lp3, mean, var = sh.logpdf_and_posterior_marginals(y_nan[lo:hi], np.array([0.05]))
            out[rank] = (lp, lp3, lo, hi, np.asarray(mean), np.asarray(var))
        except Exception as ex:          # noqa: BLE001
            errs.append(ex)
            barrier.abort()
    for n in (lib.tgp_shard_slot_size(0, d), lib.tgp_shard_slot_size(1, d), lib.tgp_shard_steady_slot_size(d), 1):
        shared[("g", n)] = torch.zeros(world * n, dtype=torch.float64, device="cuda:0")
    for hh in range(16, 1537, 16):
        shared[("g", 2 * hh)] = torch.zeros(world * 2 * hh, dtype=torch.float64, device="cuda:0")
    shared[("r", 4)] = torch.zeros(world, 4, dtype=torch.float64, device="cuda:0")
    shared[("r", 1)] = torch.zeros(world, 1, dtype=torch.float64, device="cuda:0")
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank hangs in a collective the others never entered"
    assert not errs, errs
    mean, var = np.zeros(T), np.zeros(T)
    for r in range(world):
        lp, lp3, lo, hi, m, v = out[r]
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref) and abs(lp3 - lp_ref) <= 1e-10 * abs(lp_ref)
        mean[lo:hi], var[lo:hi] = m, v
    assert np.max(np.abs(mean - pm)) <= 1e-8 and np.max(np.abs(var - pv)) <= 1e-8
