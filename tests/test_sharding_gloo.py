"""CPU tier: the time-sharded (multi-GPU) path of temporalgps.jl_amd/parallel.py, world_size 2 and 3 over gloo.
The product's ShardedLGSSM host logic and the library's host-side monoid functions (tgp_elem_apply) run
for real; only the per-segment device work is replaced by the CPU emulation of the same kernels' math
(tests/hostsim) through the injectable `engine` -- there is no GPU in this tier."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import components as oc
from oracle import lgssm_ref as ref
from tests import _util as U


def _slice_model(model, lo, hi):
    out = dict(model, T=hi - lo)
    for k in ("A", "a", "Q", "H", "h", "R"):
        arr = np.atleast_1d(model[k])
        out[k] = arr[lo:hi] if arr.shape[0] > 1 else arr
    return out


class SimEngine:
    """Segment engine backed by tests/hostsim (same chunk/monoid code as the HIP kernels, run on the host)."""

    def __init__(self, seg_model, lib):
        self.m, self.lib, self.d = seg_model, lib, len(seg_model["x0m"])
        self.carry = (seg_model["x0m"], seg_model["x0P"])

    def x0(self):
        return self.m["x0m"].copy(), self.m["x0P"].copy()

    def segment_reduce(self, y):
        return U.hostsim_run(self.m, 5, y=y, want_elem=True)["elem"]

    def elem_apply(self, kind, elem, m, P):
        mo, Po = np.empty(self.d), np.empty((self.d, self.d))
        Pc = np.ascontiguousarray(np.asarray(P).T)
        e, mc = np.ascontiguousarray(elem), np.ascontiguousarray(m)
        assert self.lib.tgp_elem_apply(kind, self.d, e.ctypes.data, mc.ctypes.data, Pc.ctypes.data, mo.ctypes.data, Po.ctypes.data) == 0
        return mo, Po.T.copy()

    def set_x0(self, m, P):
        self.carry = (np.array(m), np.array(P))

    def _seg(self):
        return dict(self.m, x0m=self.carry[0], x0P=self.carry[1])

    def logpdf(self, y, reuse):
        return U.hostsim_run(self._seg(), 0, y=y)["lml"]

    def smoother_forward(self, y, reuse):
        r = U.hostsim_run(self._seg(), 2, y=y, want_rev=True)
        return r["rev"], r["xfm"], r["xfP"], r["lml"]

    def smoother_backward(self, xs, R_new, like):
        r = U.hostsim_run(self._seg(), 2, y=self._y, Rnew=R_new, xs=xs)
        return r["mean"], r["var"]


def _worker(rank, world, port, case, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import temporalgps_jl_amd as tgp
        from temporalgps_jl_amd import parallel
        model, y = case
        lo, hi = parallel.segment_bounds(model["T"], world, rank)
        eng = SimEngine(_slice_model(model, lo, hi), tgp._lib.load())
        eng._y = y[lo:hi]
        sh = parallel.ShardedLGSSM(None, world, rank, engine=eng)
        lp = sh.logpdf(y[lo:hi])
        Rn = np.full(hi - lo, 0.05)
        mean, var = sh.posterior_marginals(y[lo:hi], Rn)
        ret[rank] = (lp, lo, hi, mean, var)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind", ["lti", "tv"])
def test_time_sharded_equals_sequential(world, kind):
    rng = np.random.default_rng(7)
    if kind == "lti":
        model, y, _ = U.gp_case(("matern52",), ("regular", 0.0, 0.1, 157), 0.1, seed=3)
    else:
        model = U.random_lgssm(rng, True, 3, 101)
        y = rng.standard_normal(101)
    T = model["T"]
    lp_ref = ref.logpdf(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior(model, y), np.full(T, 0.05)))
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, (model, y), ret), nprocs=world, join=True)
    assert len(ret) == world
    mean, var = np.zeros(T), np.zeros(T)
    for r in range(world):
        lp, lo, hi, m, v = ret[r]
        assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref)            # all_reduce(sum): every rank has the total
        mean[lo:hi], var[lo:hi] = m, v
    np.testing.assert_allclose(mean, pm, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(var, pv, rtol=1e-8, atol=1e-9)


def test_segment_bounds_cover_and_balance():
    from temporalgps_jl_amd import parallel
    for T, W in [(10, 3), (10_000_000, 8), (7, 7), (100_000_001, 8)]:
        segs = [parallel.segment_bounds(T, W, r) for r in range(W)]
        assert segs[0][0] == 0 and segs[-1][1] == T
        assert all(segs[i][1] == segs[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in segs]
        # balanced to one step for short series; long ones put interior boundaries on multiples of 512 steps (whole tiles of the
        # stationary-gain engine for every segment that hands its end state on)
        if T // W >= 8 * 512:
            assert all(a % 512 == 0 for a, _ in segs) and max(sizes) - min(sizes) <= 2 * 512
        else:
            assert max(sizes) - min(sizes) <= 1
    # the in-library handle splits the same way (tgp_multi_segment)
    from temporalgps_jl_amd import multi
    for T, W in [(10, 3), (10_000_000, 8), (100_000_001, 8), (9000, 3)]:
        assert [multi.segment_bounds(T, W, r) for r in range(W)] == [parallel.segment_bounds(T, W, r) for r in range(W)]
