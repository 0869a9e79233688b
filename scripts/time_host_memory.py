#!/usr/bin/env python3
"""End-to-end calls from PAGEABLE host memory (what a Julia Vector{Float64} / NumPy caller hands over): y uploaded over PCIe, (mean, var)
returned to the host. usage: time_host_memory.py [T]   (TGP_STAGING=0: plain hipMemcpyAsync; TGP_COPY_THREADS=n)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import temporalgps_jl_amd as tgp  # noqa: E402

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
model = bench.build_model(tgp, "matern52_d3", T, "lti", 0)
y = np.random.default_rng(0).standard_normal(T)
rn = np.array([1e-18])
mean, var = np.empty(T), np.empty(T)


def timed(fn, n=5):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


t_lp = timed(lambda: tgp.logpdf(model, y))
t_pm = timed(lambda: tgp.posterior_marginals(model, y, rn))                       # fresh output arrays each call (first-touch page faults included)
t_pm2 = timed(lambda: tgp.posterior_marginals(model, y, rn, out=(mean, var)))     # outputs written into arrays that already have their pages
print(f"TGP_STAGING={os.environ.get('TGP_STAGING', '1')} threads={os.environ.get('TGP_COPY_THREADS', '8')} T={T}: logpdf {t_lp * 1e3:.2f} ms "
      f"({8 * T / t_lp / 1e9:.1f} GB/s in) | posterior marginals {t_pm * 1e3:.2f} ms, into reused outputs {t_pm2 * 1e3:.2f} ms "
      f"({24 * T / t_pm2 / 1e9:.1f} GB/s in+out) | logpdf + posterior marginals {T / (t_lp + t_pm2):.3e} steps/s end to end")
