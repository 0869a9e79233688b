"""Where the Python mirror's per-call time goes (cfg2 call path): each piece of lgssm.logpdf / posterior_marginals timed alone."""
import ctypes
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib as L
from temporalgps_jl_amd import lgssm as G
from temporalgps_jl_amd import lti_sde as P

import os
_getcpu = ctypes.CDLL(None).sched_getcpu
print("started on cpu", _getcpu(), "of", len(os.sched_getaffinity(0)))
if os.environ.get("BIND") == "1":
    print("bound:", L.bind_host_thread(0), "now on cpu", _getcpu(), "of", len(os.sched_getaffinity(0)))
if os.environ.get("PIN"):
    os.sched_setaffinity(0, {int(os.environ["PIN"])})
    print("pinned to cpu", _getcpu())
T = 10_000_000
model = P.build_lgssm(P.to_kernel(("matern52",)), P.RegularSpacing(0.0, 0.1, T), 0.1)
hd = model.handle()
y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
mean, var = torch.empty_like(y), torch.empty_like(y)
tgp.logpdf(model, y)
torch.cuda.synchronize()


def t(name, fn, n=20000):
    for _ in range(100):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"{name:60s} {(time.perf_counter() - t0) / n * 1e6:7.2f} us")


t("empty lambda", lambda: None)
t("torch.cuda.current_stream(dev)", lambda: torch.cuda.current_stream(y.device))
s = torch.cuda.current_stream(y.device)
t("stream.synchronize()", lambda: s.synchronize())
t("stream.query()", lambda: s.query())
t("torch._C._cuda_getCurrentRawStream(0)", lambda: torch._C._cuda_getCurrentRawStream(0))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
hip.hipStreamQuery.argtypes = [ctypes.c_void_p]
raw = torch._C._cuda_getCurrentRawStream(0)
print("raw stream", raw)
t("ctypes hipStreamSynchronize(raw)", lambda: hip.hipStreamSynchronize(raw))
t("ctypes hipStreamQuery(raw)", lambda: hip.hipStreamQuery(raw))
t("G._sync_torch(y)", lambda: G._sync_torch(y))
t("y.to(float64).contiguous()", lambda: y.to(torch.float64).contiguous())
t("G._obs(y, model, lazy_nan=True)", lambda: G._obs(y, model, lazy_nan=True))
t("G._check_inputs", lambda: G._check_inputs(model, y))
t("model.handle()", lambda: model.handle())
t("L.ptr(y)", lambda: L.ptr(y))
t("isinstance(model, PosteriorLGSSM)", lambda: isinstance(model, G.PosteriorLGSSM))
out = ctypes.c_double()
t("ctypes.c_double() + byref", lambda: ctypes.byref(ctypes.c_double()))
t("tgp_last_error (a trivial ctypes call)", lambda: hd.lib.tgp_last_error(hd.h))
t("python logpdf", lambda: tgp.logpdf(model, y), 2000)
yp = L.ptr(y)
t("C logpdf", lambda: hd.lib.tgp_logpdf(hd.h, yp, None, L.IN_DEVICE, ctypes.byref(out)), 2000)
t("python posterior_marginals(out=)", lambda: tgp.posterior_marginals(model, y, Rnew, out=(mean, var)), 2000)
rp, mp, vp = L.ptr(Rnew), L.ptr(mean), L.ptr(var)
fl = L.IN_DEVICE | L.OUT_DEVICE | L.SHARED_R
t("C posterior_marginals", lambda: hd.lib.tgp_posterior_marginals(hd.h, yp, None, rp, fl, mp, vp, None), 2000)
print("ends on cpu", _getcpu())
t("C fused", lambda: hd.lib.tgp_logpdf_and_posterior_marginals(hd.h, yp, None, rp, fl, ctypes.byref(out), mp, vp), 2000)
