"""The reference's prediction call chain end to end through the Python mirror (posterior_lti_sde.jl:18-36 -> merge_datasets :97-123 -> the LGSSM of the
merged inputs -> marginals): where the wall clock of `mean_and_var(posterior(fx, y)(x_new))` goes.  usage: time_predict_path.py [T] [n_new]"""
import cProfile
import pstats
import sys
import time

import numpy as np

from temporalgps_jl_amd import lti_sde as P

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n_new = int(float(sys.argv[2])) if len(sys.argv) > 2 else T // 10
rng = np.random.default_rng(0)
x = P.RegularSpacing(0.0, 0.1, T)
f = P.to_sde(P.GP(P.Matern52Kernel()), P.HIPStorage(device=0))
fx = f(x, 0.1)
y = rng.standard_normal(T)
x_new = np.sort(rng.uniform(0.0, 0.1 * T, n_new))
post = P.posterior(fx, y)
for _ in range(2):
    m, v = P.mean_and_var(post(x_new))
t0 = time.perf_counter()
n = 3
for _ in range(n):
    m, v = P.mean_and_var(post(x_new))
dt = (time.perf_counter() - t0) / n
print(f"T = {T}, {n_new} new inputs: mean_and_var(posterior(fx, y)(x_new)) {dt * 1e3:.1f} ms per call")
pr = cProfile.Profile()
pr.enable()
P.mean_and_var(post(x_new))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
