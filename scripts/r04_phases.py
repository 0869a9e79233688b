"""Phases of the one-launch kernel's head wave per workload (TGP_STEADY_DEBUG prints them from the library)."""
import os, sys
os.environ["TGP_STEADY_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import temporalgps_jl_amd as tgp
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for name in ["matern52_d3", "matern32_d2", "sum52_12_d4", "sum52_32_d5", "sum52_52s_d6", "sum52_32s_32_d7", "sum52_52s_32_d8"]:
    model = bench.build_model(tgp, name, T, "lti", 0)
    y = torch.randn(T, dtype=torch.float64, device="cuda")
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda")
    print(name, flush=True)
    for _ in range(3):
        tgp.logpdf_and_posterior_marginals(model, y, Rn)
