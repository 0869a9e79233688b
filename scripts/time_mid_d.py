"""State dimensions 17..64 on the dense engine (one kernel chain per time step): steps/s of logpdf and of logpdf + posterior
marginals, next to a NumPy sequential filter on the host (the reference's ArrayStorage path is BLAS calls of this size)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import temporalgps_jl_amd as tgp
from tests import _util as U
from tests.test_gpu_parity import to_device_model

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for d in (17, 24, 32, 48, 64):
    rng = np.random.default_rng(d)
    model = U.random_lgssm(rng, False, d, T)
    y = rng.standard_normal(T)
    dm = to_device_model(tgp, model)
    tgp.logpdf(dm, y)
    t0 = time.perf_counter(); lp = tgp.logpdf(dm, y); t1 = time.perf_counter()
    Rn = np.full(1, 0.1)
    tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    t2 = time.perf_counter(); tgp.logpdf_and_posterior_marginals(dm, y, Rn); t3 = time.perf_counter()
    # host: plain NumPy Kalman filter, 2000 steps
    A, a, Q, H, h, R = model["A"][0], model["a"][0], model["Q"][0], model["H"][0], float(np.ravel(model["h"])[0]), float(np.ravel(model["R"])[0])
    m, P = model["x0m"].copy(), model["x0P"].copy()
    n = 2000
    c0 = time.perf_counter()
    for t in range(n):
        m = A @ m + a; P = A @ P @ A.T + Q
        v = P @ H; s = H @ v + R; k = v / s
        m = m + k * (y[t] - H @ m - h); P = P - np.outer(k, v)
    c1 = time.perf_counter()
    print(f"d={d}: logpdf {1e6 * (t1 - t0) / T:.1f} us/step ({T / (t1 - t0):.3e} steps/s), logpdf+posterior marginals {1e6 * (t3 - t2) / T:.1f} us/step; "
          f"NumPy host filter {1e6 * (c1 - c0) / n:.1f} us/step", flush=True)
