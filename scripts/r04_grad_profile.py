"""Development: where the time of logpdf_and_gradient (adjoint method) goes -- kernels (hipEvent profile) against the host's share (cProfile)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P

T = 10_000_000
k = P.ScaledKernel(1.3, P.StretchedKernel(0.9, P.Matern52Kernel()))
fx = P.to_sde(P.GP(k))(P.RegularSpacing(0.0, 0.01, T), 0.1)
y = torch.randn(T, dtype=torch.float64, device="cuda:0")
for _ in range(5):
    lp, g = P.logpdf_and_gradient(fx, y, method="adjoint")
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 100
for _ in range(N):
    lp, g = P.logpdf_and_gradient(fx, y, method="adjoint")
torch.cuda.synchronize()
print("ms per evaluation:", (time.perf_counter() - t0) / N * 1e3, g)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    lp, g = P.logpdf_and_gradient(fx, y, method="adjoint")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
