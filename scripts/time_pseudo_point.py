"""Pseudo-point (DTC / ELBO) on a long space-time series: M = 5 pseudo-points x Matern-5/2 in time = state dimension 15,
N = 20 observations per time step; sixteen-lanes-per-chunk kernels (default) against the out-of-line build."""
import sys, time, gc
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as S, space_time as ST, pseudo_point as pp, lgssm as L, _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(0)
N, M = 20, 5
r, z = rng.standard_normal(N), np.linspace(-1.5, 1.5, M)
k = 0.8 * ST.Separable(ST.SEKernel(), S.Matern52Kernel())
grid = ST.RectilinearGrid(r, S.RegularSpacing(0.0, 0.1, T))
y = rng.standard_normal(T * N)
for grp in (1, 0):
    model = pp.build_lgssm(k, grid, z, 0.2)
    hd = model.handle()
    hd.set_option(_lib.OPT_GROUP, grp)
    Y = y.reshape(T, N)
    L.logpdf(model, Y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lp = L.logpdf(model, Y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"RESULT group={grp} state dim {model.dim} T={T} N={N}: dtc (logpdf) {lp:.4f} in {(t1-t0)*1e3:.1f} ms", flush=True)
    model = hd = None; gc.collect(); torch.cuda.synchronize()
