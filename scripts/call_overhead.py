"""Where a headline call's time goes outside its kernel (cfg2: Matern-5/2, T = 1e7 unless told otherwise): wall clock of the Python call, of the
bare C call (ctypes, device pointers, outputs reused), the kernel's own duration (hipEvents), and the library's own phase stamps
(TGP_STEADY_DEBUG=1 prints them)."""
import ctypes
import sys
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib as L
from temporalgps_jl_amd import lti_sde as P

import os
if os.environ.get("BIND", "1") != "0":      # (as bench.py does: the process on the GPU's socket; BIND=0 for the A/B)
    L.bind_host_thread(0)
T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kname = sys.argv[2] if len(sys.argv) > 2 else "matern52"
dev = "cuda:0"
spec = eval(kname) if kname.startswith("(") else (kname,)      # a kernel expression as bench.py writes them, e.g. '("sum", ("matern52",), ("stretched", 2.0, ("matern52",)))'
model = P.build_lgssm(P.to_kernel(spec), P.RegularSpacing(0.0, 0.1, T), 0.1)
hd = model.handle()
y = torch.randn((T,), dtype=torch.float64, device=dev)
Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device=dev)
mean, var = torch.empty_like(y), torch.empty_like(y)
N = 200


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


out = ctypes.c_double()
flags_lp = L.IN_DEVICE
flags_pm = L.IN_DEVICE | L.OUT_DEVICE | L.SHARED_R
yp, rp, mp, vp = L.ptr(y), L.ptr(Rnew), L.ptr(mean), L.ptr(var)
res = {}
res["python logpdf"] = timed(lambda: tgp.logpdf(model, y))
res["python posterior_marginals"] = timed(lambda: tgp.posterior_marginals(model, y, Rnew))
res["python posterior_marginals(out=)"] = timed(lambda: tgp.posterior_marginals(model, y, Rnew, out=(mean, var)))
res["python fused(out=)"] = timed(lambda: tgp.logpdf_and_posterior_marginals(model, y, Rnew, out=(mean, var)))
res["C logpdf"] = timed(lambda: hd.lib.tgp_logpdf(hd.h, yp, None, flags_lp, ctypes.byref(out)))
res["C posterior_marginals"] = timed(lambda: hd.lib.tgp_posterior_marginals(hd.h, yp, None, rp, flags_pm, mp, vp, None))
res["C fused"] = timed(lambda: hd.lib.tgp_logpdf_and_posterior_marginals(hd.h, yp, None, rp, flags_pm, ctypes.byref(out), mp, vp))
res["torch current_stream().synchronize()"] = timed(lambda: torch.cuda.current_stream().synchronize())
res["torch.empty x2"] = timed(lambda: (torch.empty_like(y), torch.empty_like(y)))
hd.set_option(L.OPT_PROFILE, 1)
hd.profile_reset()
for _ in range(20):
    hd.lib.tgp_logpdf(hd.h, yp, None, flags_lp, ctypes.byref(out))
    hd.lib.tgp_posterior_marginals(hd.h, yp, None, rp, flags_pm, mp, vp, None)
hd.set_option(L.OPT_PROFILE, 0)
for k, v in hd.profile().items():
    res["kernel " + k] = v["total_ms"] / v["calls"] * 1e3
for k, v in res.items():
    print(f"{k:44s} {v:8.1f} us")
