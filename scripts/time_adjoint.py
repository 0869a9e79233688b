#!/usr/bin/env python3
"""logpdf + gradient on the device: ONE adjoint pass (tgp_logpdf_adjoint) against the forward-mode tangent scans, T = 1e7.
usage: time_adjoint.py [T]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import temporalgps_jl_amd as tgp  # noqa: E402
from temporalgps_jl_amd import lti_sde as P  # noqa: E402

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
CASES = {"matern52 (3 parameters, d=3)": lambda: P.GP(P.ScaledKernel(1.0, P.StretchedKernel(1.0, P.Matern52Kernel()))),
         "52+32+12 with a constant mean (8 parameters, d=6)": lambda: P.GP(P.ConstMean(0.3), P.ScaledKernel(1.0, P.StretchedKernel(1.0, P.Matern52Kernel()))
                                                                         + P.ScaledKernel(0.5, P.StretchedKernel(1.5, P.Matern32Kernel()))
                                                                         + P.ScaledKernel(0.3, P.StretchedKernel(0.7, P.Matern12Kernel())))}
y = torch.randn(T, dtype=torch.float64, device="cuda:0")
for name, mk in CASES.items():
    fx = P.to_sde(mk(), P.HIPStorage(device=0))(P.RegularSpacing(0.0, 0.1, T), 0.1)

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out
    t_lp, lp = timed(lambda: P.logpdf(fx, y))
    t_a, (lpa, ga) = timed(lambda: P.logpdf_and_gradient(fx, y, method="adjoint"))
    t_t, (lpt, gt) = timed(lambda: P.logpdf_and_gradient(fx, y, method="tangent"), 2)
    sc = max(abs(v) for v in gt.values())
    print(f"{name}, T = {T}: logpdf {t_lp * 1e3:.3f} ms | adjoint {t_a * 1e3:.3f} ms ({t_a / t_lp:.2f} x logpdf, {T / t_a:.3e} steps/s) | "
          f"tangent scans {t_t * 1e3:.3f} ms ({t_t / t_lp:.2f} x) | max |adjoint - tangent| / max |g| = {max(abs(ga[k] - gt[k]) for k in gt) / sc:.2e}")
    hd = fx.build_lgssm().handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    P.logpdf_and_gradient(fx, y, method="adjoint")
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    print("   ", {k: round(v["total_ms"] / max(1, v["calls"]) * 1e3, 1) for k, v in hd.profile().items()})
