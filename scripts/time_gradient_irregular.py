"""logpdf + gradient on irregularly spaced inputs (tgp_logpdf_grad_sde): timing at T = 2e6, Matern-5/2."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import temporalgps_jl_amd as tgp  # noqa: F401
from temporalgps_jl_amd import lti_sde as S
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(0)
t = np.sort(rng.uniform(0.0, 0.05 * T, T)) + np.arange(T) * 1e-6
y = rng.standard_normal(T)
fx = S.to_sde(S.GP(1.3 * S.Matern52Kernel().stretch(0.9)))(t, 0.25)
for _ in range(2):
    lp, g = S.logpdf_and_gradient(fx, y)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); lp, g = S.logpdf_and_gradient(fx, y); ts.append(time.perf_counter() - t0)
print(f"RESULT T={T} irregular: logpdf + {len(g)} derivatives in {min(ts)*1e3:.1f} ms (incl. host upload of y and the time stamps); lp={lp:.6f} grad={g}")
