"""Development: the bare C call of tgp_logpdf_adjoint (one-launch form against the five-launch form, TGP_OPT_STEADY = 2) and its kernels."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib

T = 10_000_000
for name in (sys.argv[1:] or ["matern52_d3"]):
    for opt in (3, 2):
        model = bench.build_model(tgp, name, T, "lti", 0)
        model.handle_options[tgp._lib.OPT_STEADY] = opt
        hd = model.handle()
        d = model.dim
        y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
        gA, ga, gQ, gH, gx0m, gx0P = np.zeros((d, d)), np.zeros(d), np.zeros((d, d)), np.zeros(d), np.zeros(d), np.zeros((d, d))
        ghh, gR, lml = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        args = (hd.h, _lib.ptr(y), _lib.IN_DEVICE, ctypes.byref(lml), _lib.ptr(gA), _lib.ptr(ga), _lib.ptr(gQ), _lib.ptr(gH), ctypes.byref(ghh), ctypes.byref(gR), _lib.ptr(gx0m), _lib.ptr(gx0P))
        f = hd.lib.tgp_logpdf_adjoint
        for _ in range(5):
            hd.check(f(*args))
        torch.cuda.synchronize()
        N = 50
        t0 = time.perf_counter()
        for _ in range(N):
            f(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        hd.profile_reset()
        for _ in range(5):
            f(*args)
        hd.set_option(tgp._lib.OPT_PROFILE, 0)
        prof = {k: round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in hd.profile().items()}
        print(f"{name} option {opt}: tgp_logpdf_adjoint {dt * 1e3:.4f} ms  lml {lml.value:.6f} gR {gR.value:.6e}  kernels(us) {prof}", flush=True)
        del model
