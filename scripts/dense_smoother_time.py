"""Time of the dense (d = 768, p = 256) filter + RTS smoother (tgp_logpdf_and_posterior_marginals) per time step."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib, lti_sde, space_time

Nr = 256
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = np.linspace(-3.0, 3.0, Nr)
k = space_time.Separable(space_time.SEKernel(), lti_sde.to_kernel(("matern52",)))
grid = space_time.RectilinearGrid(r, lti_sde.RegularSpacing(0.0, 0.01, T))
Y = np.random.default_rng(0).standard_normal((T, Nr))
for structure in (1, 0):
    dm = space_time.build_lgssm(k, grid, 0.1)
    dm.handle_options[_lib.OPT_DENSE_STRUCTURE] = structure
    if seg:
        dm.handle_options[_lib.OPT_CHUNK] = seg
    tgp.logpdf(dm, Y[:T])
    t0 = time.perf_counter(); lp = tgp.logpdf(dm, Y); t1 = time.perf_counter()
    out = tgp.logpdf_and_posterior_marginals(dm, Y, np.full((1, Nr), 0.1)); t2 = time.perf_counter()
    out = tgp.logpdf_and_posterior_marginals(dm, Y, np.full((1, Nr), 0.1)); t3 = time.perf_counter()
    print(f"structure={structure} T={T} seg={seg}: filter {1e6 * (t1 - t0) / T:.1f} us/step, filter+smoother {1e6 * (t3 - t2) / T:.1f} us/step, "
          f"lml diff {abs(out[0] - lp):.2e}", flush=True)
    hd = dm.handle()
    hd.set_option(_lib.OPT_PROFILE, 1)
    tgp.logpdf_and_posterior_marginals(dm, Y, np.full((1, Nr), 0.1))
    for name, v in sorted(hd.profile().items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"   {name:45s} {1e3 * v['total_ms'] / max(v['calls'], 1):9.1f} us x {v['calls']}")
    hd.set_option(_lib.OPT_PROFILE, 0)
