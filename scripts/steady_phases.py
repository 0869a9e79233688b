#!/usr/bin/env python3
"""phases of k_setup_core / setup_side (TGP_STEADY_DEBUG=1 prints them on tgp_steady_steps) per workload. usage: steady_phases.py [T]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TGP_STEADY_DEBUG"] = "1"
import torch  # noqa: E402

import bench  # noqa: E402
import temporalgps_jl_amd as tgp  # noqa: E402

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
y = torch.randn(T, dtype=torch.float64, device="cuda:0")
Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
for name in bench.WORKLOADS:
    model = bench.build_model(tgp, name, T, "lti", 0)
    hd = model.handle()
    for _ in range(2):
        tgp.logpdf_and_posterior_marginals(model, y, Rn)
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    print(name, flush=True)
    hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(a), ctypes.byref(b)))
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    for _ in range(3):
        tgp.logpdf_and_posterior_marginals(model, y, Rn)
    print("   ", {k: round(v["total_ms"] / max(1, v["calls"]) * 1e3, 1) for k, v in hd.profile().items()}, flush=True)
