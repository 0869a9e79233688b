import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, _lib
SPECS = {7: ("sum", ("matern52",), ("matern32",), ("matern32",)), 8: ("sum", ("matern52",), ("matern52",), ("matern32",))}
T = 10_000_000
for d in (8,):
    model = lti_sde.build_lgssm(lti_sde.to_kernel(SPECS[d]), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
    hd = model.handle()
    yd = torch.randn(T, dtype=torch.float64, device="cuda:0")
    Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    hd.set_option(_lib.OPT_GROUP, 2)
    for chunk in (306, 611):
        hd.set_option(_lib.OPT_CHUNK, chunk)
        for _ in range(2): tgp.posterior_marginals(model, yd, Rn)
        hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset()
        for _ in range(2): tgp.posterior_marginals(model, yd, Rn)
        prof = hd.profile(); hd.set_option(_lib.OPT_PROFILE, 0)
        tot = sum(v["total_ms"] / v["calls"] for v in prof.values())
        print(f"RESULT d={d} chunk={chunk} posterior kernels {tot:.2f} ms | " + " ".join(f"{k.replace('k_','')}={v['total_ms']/v['calls']*1e3:.0f}" for k, v in prof.items()), flush=True)
