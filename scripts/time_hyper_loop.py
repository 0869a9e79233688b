"""A hyper-parameter optimisation loop as the reference's examples run it (examples/exact_time_learning.jl): every iteration builds the GP with NEW
hyper-parameters and evaluates logpdf + gradient on the same device-resident series.  Wall clock per iteration against the evaluation alone on a
model that is kept.  usage: time_hyper_loop.py [T]"""
import sys
import time

import numpy as np
import torch

from temporalgps_jl_amd import lti_sde as P

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
y = torch.randn((T,), dtype=torch.float64, device="cuda")
x = P.RegularSpacing(0.0, 0.1, T)


def make(theta):
    k = P.ScaledKernel(float(np.exp(theta[0])), P.StretchedKernel(float(np.exp(theta[1])), P.Matern52Kernel()))
    return P.to_sde(P.GP(k), P.HIPStorage(device=0))(x, float(np.exp(theta[2])))


theta = np.array([0.0, 0.0, np.log(0.1)])
fx = make(theta)
for _ in range(3):
    P.logpdf_and_gradient(fx, y)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    P.logpdf_and_gradient(fx, y)
t_keep = (time.perf_counter() - t0) / n
rng = np.random.default_rng(0)
t0 = time.perf_counter()
for i in range(n):
    th = theta + 0.01 * rng.standard_normal(3)
    fx = make(th)
    lp, g = P.logpdf_and_gradient(fx, y)
t_new = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for i in range(n):
    th = theta + 0.01 * rng.standard_normal(3)
    fx = make(th)
    lp = P.logpdf(fx, y)
t_lp = (time.perf_counter() - t0) / n
print(f"T = {T}: logpdf + gradient on a kept model {t_keep * 1e3:.3f} ms; with a NEW model every iteration {t_new * 1e3:.3f} ms ({T / t_new:.3e} steps/s); "
      f"logpdf alone with a new model every iteration {t_lp * 1e3:.3f} ms")
