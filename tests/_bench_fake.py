"""TEST INFRASTRUCTURE: per-segment engine for `bench.py --engine-factory tests._bench_fake:make_engine` -- the host emulation
of the chunk kernels (tests/hostsim, the product's own headers run on the CPU) behind the ShardedLGSSM engine interface, so
that the CPU tier can run bench.py's N > 1 launch / sharding / collective path with gloo on a box without GPUs."""
import numpy as np

from oracle import components as oc
from tests.test_sharding_gloo import SimEngine, _slice_model


def make_engine(workload, T, seg):
    import bench
    import temporalgps_jl_amd as tgp
    k, d, dt, s2 = bench.WORKLOADS[workload]
    model = oc.build_lgssm(k, ("regular", 0.0, dt, T), s2)
    y = np.random.default_rng(11).standard_normal(T)
    lo, hi = seg
    eng = SimEngine(_slice_model(model, lo, hi), tgp._lib.load())
    eng._y = y[lo:hi]
    return eng, y[lo:hi], np.full(hi - lo, 1e-18)
