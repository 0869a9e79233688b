"""16 < d <= 64: prior marginals and rand, persistent kernels against the per-step kernel chain (TGP_OPT_DENSE_FUSED = 0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib
from tests import _util as U
from tests.test_gpu_parity import to_device_model
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for d in (24, 64):
    rng = np.random.default_rng(d)
    model = U.random_lgssm(rng, False, d, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    row = []
    for fused in (1, 0):
        dm = to_device_model(tgp, model)
        dm.handle_options[_lib.OPT_DENSE_FUSED] = fused
        tgp.marginals(dm); tgp.rand(eps, dm)
        t0 = time.perf_counter(); tgp.marginals(dm); t1 = time.perf_counter(); tgp.rand(eps, dm); t2 = time.perf_counter()
        row.append(f"{'persistent' if fused else 'chain'}: marginals {1e6 * (t1 - t0) / T:.1f} us/step, rand {1e6 * (t2 - t1) / T:.1f} us/step")
    print(f"d={d} T={T}: " + "; ".join(row), flush=True)
