"""Development: the binding's share of a combined call -- the Python path against the bare C entry point with prepared arguments."""
import ctypes, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib

T = 10_000_000
model = bench.build_model(tgp, "matern52_d3", T, "lti", 0)
hd = model.handle()
y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
for _ in range(5):
    tgp.logpdf_and_posterior_marginals(model, y, Rn)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    tgp.logpdf_and_posterior_marginals(model, y, Rn)
torch.cuda.synchronize()
t_py = (time.perf_counter() - t0) / N
mean, var = torch.empty_like(y), torch.empty_like(y)
lml = ctypes.c_double()
args = (hd.h, _lib.ptr(y), None, _lib.ptr(Rn), _lib.IN_DEVICE | _lib.OUT_DEVICE | _lib.SHARED_R, ctypes.byref(lml), _lib.ptr(mean), _lib.ptr(var))
f = hd.lib.tgp_logpdf_and_posterior_marginals
for _ in range(5):
    f(*args)
t0 = time.perf_counter()
for _ in range(N):
    f(*args)
torch.cuda.synchronize()
t_c = (time.perf_counter() - t0) / N
print(f"python path {t_py * 1e6:.1f} us per call, bare C entry point {t_c * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tgp.logpdf_and_posterior_marginals(model, y, Rn)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
