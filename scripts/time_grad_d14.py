"""logpdf + gradient for ApproxPeriodicKernel (d = 14) and a d = 9 sum kernel: forward-mode tangent scans (out-of-line dual-number
kernels for d >= 9) against central differences of the (group-kernel) logpdf."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for name, k in (("approx_periodic d=14", 1.3 * P.ApproxPeriodicKernel().stretch(0.8)), ("3 x matern52 d=9", P.Matern52Kernel() + P.Matern52Kernel().stretch(0.5) + 0.5 * P.Matern52Kernel().stretch(2.0))):
    fx = P.to_sde(P.GP(k))(P.RegularSpacing(0.0, 0.05, T), 0.2)
    y = np.random.default_rng(0).standard_normal(T)
    P.logpdf(fx, y)
    t0 = time.perf_counter(); lp = P.logpdf(fx, y); t1 = time.perf_counter()
    try:
        P.logpdf_and_gradient(fx, y)
        t2 = time.perf_counter(); lp2, g = P.logpdf_and_gradient(fx, y); t3 = time.perf_counter()
        print(f"{name} T={T}: logpdf {1e3 * (t1 - t0):.1f} ms, logpdf_and_gradient ({len(g)} parameters) {1e3 * (t3 - t2):.1f} ms", flush=True)
    except Exception as ex:
        print(name, "gradient failed:", repr(ex)[:200])
