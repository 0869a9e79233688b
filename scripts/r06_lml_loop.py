"""Round 6: a bare loop of headline calls for the profilers: python scripts/r06_lml_loop.py [logpdf|post|both] [n] [d]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib as L
from temporalgps_jl_amd import lti_sde as P

what = sys.argv[1] if len(sys.argv) > 1 else "logpdf"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kern = {3: ("matern52",), 2: ("matern32",), 1: ("matern12",)}[int(sys.argv[3]) if len(sys.argv) > 3 else 3]
T = 10_000_000
model = P.build_lgssm(P.to_kernel(kern), P.RegularSpacing(0.0, 0.1, T), 0.1)
hd = model.handle()
y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
mean, var = torch.empty_like(y), torch.empty_like(y)
out = ctypes.c_double()
yp, rp, mp, vp = L.ptr(y), L.ptr(Rnew), L.ptr(mean), L.ptr(var)
torch.cuda.synchronize()
for _ in range(n):
    if what in ("logpdf", "both"):
        hd.lib.tgp_logpdf(hd.h, yp, None, L.IN_DEVICE, ctypes.byref(out))
    if what in ("post", "both"):
        hd.lib.tgp_posterior_marginals(hd.h, yp, None, rp, L.IN_DEVICE | L.OUT_DEVICE | L.SHARED_R, mp, vp, None)
print(out.value)
