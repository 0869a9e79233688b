#!/usr/bin/env python3
"""Irregularly spaced inputs (tgp_model_set_sde: the device builds A_k = exp(F dt_k), Q_k from the time stamps): one combined
logpdf + posterior-marginals call and one logpdf call at T = 1e7, dt ~ U(0.05, 0.15), beside the same kernel on a regular grid.
usage: time_irregular.py [T] [kernel]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import temporalgps_jl_amd as tgp  # noqa: E402
from temporalgps_jl_amd import lti_sde as P  # noqa: E402

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
KERNELS = {"matern32": lambda: P.Matern32Kernel(), "matern52": lambda: P.Matern52Kernel(),
           "sum52_12": lambda: P.Matern52Kernel() + P.ScaledKernel(0.5, P.StretchedKernel(1.5, P.Matern12Kernel())),
           "sum52_32": lambda: P.Matern52Kernel() + P.ScaledKernel(0.5, P.StretchedKernel(1.5, P.Matern32Kernel())),
           "sum52_52": lambda: P.Matern52Kernel() + P.ScaledKernel(0.5, P.StretchedKernel(1.5, P.Matern52Kernel()))}
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["matern52"]
rng = np.random.default_rng(0)
t = np.cumsum(rng.uniform(0.05, 0.15, T))
y = torch.randn(T, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(5))
Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")


def timed(fn, n=5):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name in names:
    k = P.ScaledKernel(1.0, P.StretchedKernel(1.0, KERNELS[name]()))
    t0 = time.perf_counter()
    fx = P.to_sde(P.GP(k), P.HIPStorage(device=0))(t, 0.1)
    dm = fx.build_lgssm()
    if os.environ.get("CLOSED_FORM") is not None:
        dm.handle().set_option(tgp._lib.OPT_SDE_CLOSED_FORM, int(os.environ["CLOSED_FORM"]))
    if os.environ.get("CHUNK") is not None:
        dm.handle().set_option(tgp._lib.OPT_CHUNK, int(os.environ["CHUNK"]))
    lp = tgp.logpdf(dm, y)
    t_bind = time.perf_counter() - t0
    fr = P.to_sde(P.GP(k), P.HIPStorage(device=0))(P.RegularSpacing(0.0, 0.1, T), 0.1).build_lgssm()
    t_c = timed(lambda: tgp.logpdf_and_posterior_marginals(dm, y, Rn))
    t_l = timed(lambda: tgp.logpdf(dm, y))
    t_rc = timed(lambda: tgp.logpdf_and_posterior_marginals(fr, y, Rn))
    d = dm.dim
    print(f"{name} d={d} T={T} lp={lp!r}: irregular combined call {t_c * 1e3:.3f} ms ({T / t_c:.3e} steps/s), logpdf {t_l * 1e3:.3f} ms | regular grid combined "
          f"{t_rc * 1e3:.3f} ms | ratio {t_c / t_rc:.2f} | model build + first call {t_bind:.2f} s", flush=True)
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    print("   ", {kk: round(v["total_ms"] / max(1, v["calls"]) * 1e3, 1) for kk, v in hd.profile().items()}, flush=True)
    del dm, fr, fx
