"""Wide states (16 < d <= 63): logpdf of ApproxPeriodicKernel() * Matern32Kernel() (d = 28) and friends at T = 1e6 on the stationary closed loop across the
chip (tgp_wide.hip) against the dense engine's sequential passes on one compute unit (TGP_OPT_WIDE = 0)."""
import sys
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib as L
from temporalgps_jl_amd import lti_sde as P

L.bind_host_thread(0)
T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
NODENSE = len(sys.argv) > 2 and sys.argv[2] == "nodense"      # (profile runs: the wide engine's kernels only)
SPECS = {
    18: ("product", ("approx_periodic", 3, 1.0), ("matern52",)),
    28: ("product", ("approx_periodic", 7, 1.0), ("matern32",)),
    42: ("product", ("approx_periodic", 7, 1.0), ("matern52",)),
}
for d, spec in SPECS.items():
    res = {}
    y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
    for wide in ((1,) if NODENSE else (1, 0)):
        model = P.build_lgssm(P.to_kernel(spec), P.RegularSpacing(0.0, 0.1, T), 0.1)
        model.handle_options[L.OPT_WIDE] = wide
        hd = model.handle()
        t0 = time.perf_counter()
        lp = tgp.logpdf(model, y)
        first = time.perf_counter() - t0
        n = 20 if wide else 2
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            lp = tgp.logpdf(model, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        hd.set_option(L.OPT_PROFILE, 1)
        hd.profile_reset()
        tgp.logpdf(model, y)
        prof = {k: v["total_ms"] for k, v in hd.profile().items()}
        hd.set_option(L.OPT_PROFILE, 0)
        res[wide] = (dt, first, lp, prof)
        if wide:
            Rn = torch.full((1,), 0.1, dtype=torch.float64, device="cuda:0")
            out = (torch.empty_like(y), torch.empty_like(y))
            tgp.posterior_marginals(model, y, Rn, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tgp.posterior_marginals(model, y, Rn, out=out)
            torch.cuda.synchronize()
            print(f"d={d} T={T}: wide posterior marginals {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call")
    if NODENSE:
        print(f"d={d} T={T}: wide {res[1][0] * 1e3:.3f} ms per call, kernels {res[1][3]}")
        continue
    (dw, fw, lw, pw), (dd_, fd, ld, pd) = res[1], res[0]
    print(f"d={d} T={T}: wide {dw * 1e3:.3f} ms per call (first call with the plan {fw * 1e3:.2f} ms; kernels {pw}), dense passes {dd_ * 1e3:.1f} ms "
          f"({dict(list(pd.items())[:3])}): x{dd_ / dw:.0f}; logpdf {lw:.6f} vs {ld:.6f} (rel {abs(lw - ld) / abs(ld):.1e})")
