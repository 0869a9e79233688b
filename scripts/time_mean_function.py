"""A GP with a mean function on a regular grid (per-step emission offset, everything else shared), T = 1e7, device-resident: the one-launch path
(k_smooth_one with the offset subtracted per step) against what served it before (TGP_OPT_STEADY = 2: the sweep engine for d <= 4, the general
engine beyond).  usage: time_mean_function.py [T]"""
import sys
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for spec in (("matern52",), ("sum", ("matern52",), ("stretched", 2.0, ("matern52",)))):
    res = {}
    for opt in (3, 2):
        model = P.build_lgssm(P.to_kernel(spec), P.RegularSpacing(0.0, 0.1, T), 0.1, mean=P.CustomMean(lambda v: np.sin(0.7 * v)))
        model.handle_options[tgp._lib.OPT_STEADY] = opt
        y = torch.randn((T,), dtype=torch.float64, device="cuda")
        Rn = torch.full((1,), 0.05, dtype=torch.float64, device="cuda")
        for _ in range(2):
            tgp.logpdf_and_posterior_marginals(model, y, Rn)
        torch.cuda.synchronize()
        n = 10 if opt == 3 else 3
        t0 = time.perf_counter()
        for _ in range(n):
            tgp.logpdf_and_posterior_marginals(model, y, Rn)
        torch.cuda.synchronize()
        res[opt] = (time.perf_counter() - t0) / n * 1e3
        hd = model.handle()
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        hd.profile_reset()
        tgp.logpdf_and_posterior_marginals(model, y, Rn)
        res[(opt, "k")] = sorted(hd.profile())
        del model
    print(f"d = {len(P.to_kernel(spec).to_sde()[2]) if hasattr(P.to_kernel(spec), 'to_sde') else '?'} {spec}: one launch {res[3]:.3f} ms {res[(3, 'k')]}; before {res[2]:.3f} ms {res[(2, 'k')][:4]}")
