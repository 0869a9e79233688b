"""BASELINE config 5: separable space-time GP, 256 spatial points x T = 1e5, Matern-5/2 in time, SE in space,
noise 0.1 -- through the eigen-decoupled exact shortcut (temporalgps.jl_amd/space_time.py). Prints timings."""
import sys
import time
import json
import numpy as np
sys.path.insert(0, ".")
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, space_time

Nr = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
rng = np.random.default_rng(5)
k = space_time.Separable(space_time.SEKernel(), lti_sde.Matern52Kernel())
grid = space_time.RectilinearGrid(np.linspace(-3, 3, Nr), lti_sde.RegularSpacing(0.0, 0.01, T))
t0 = time.perf_counter()
dec = space_time.DecoupledSpaceTime(k, grid, 0.1)
t1 = time.perf_counter()
y = rng.standard_normal(T * Nr)
lp = dec.logpdf(y)                       # includes model upload + tiling
t2 = time.perf_counter()
hd = dec.model.handle()
hd.set_option(tgp._lib.OPT_TIMING, 1)
res = {}
for name, fn in (("logpdf", lambda: dec.logpdf(y)), ("posterior_marginals", lambda: dec.posterior_marginals(y, 0.1))):
    fn()
    ts = []
    for _ in range(3):
        a = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - a)
    res[name] = dict(wall_s=min(ts), device_kernel_ms=hd.last_timing()["kernel_ms"])
print(json.dumps(dict(Nr=Nr, T=T, d_dense=3 * Nr, scalar_steps=Nr * T, build_host_s=t1 - t0, first_call_s=t2 - t1, logpdf_value=lp, **res,
                      dense_flop_per_logpdf=2.57e9 * T * (Nr / 256) ** 3)))
