"""rand of posterior(model, y) at T = 1e7 (device-resident series and draws): the one-launch path (tgp_posterior_rand) against the
evaluated route (tgp_posterior + tgp_rand on the Reverse model).  usage: time_posterior_rand.py [T] [kernel]"""
import sys
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kname = sys.argv[2] if len(sys.argv) > 2 else "matern52"
model = P.build_lgssm(P.to_kernel((kname,)), P.RegularSpacing(0.0, 0.1, T), 0.1)
d = model.dim
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn((T,), dtype=torch.float64, device="cuda", generator=g)
eps = (torch.randn((T, d), dtype=torch.float64, device="cuda", generator=g), torch.randn((T,), dtype=torch.float64, device="cuda", generator=g),
       np.random.default_rng(0).standard_normal(d))
Rn = np.array([0.05])


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms1, y1 = timed(lambda: tgp.rand(eps, tgp.replace_observation_noise_cov(tgp.posterior(model, y), Rn)), 20)


yh, eh = y.cpu().numpy(), (eps[0].cpu().numpy(), eps[1].cpu().numpy(), eps[2])


def evaluated():      # (the evaluated route takes host arrays: its Reverse model mixes the evaluated blocks with the prior's host emissions)
    post = tgp.replace_observation_noise_cov(tgp.posterior(model, yh), Rn)
    post.materialise()
    return tgp.rand(eh, post)


ms1h, _ = timed(lambda: tgp.rand(eh, tgp.replace_observation_noise_cov(tgp.posterior(model, yh), Rn)), 3)
ms2, y2 = timed(evaluated, 3)
hd = model.handle()
hd.set_option(tgp._lib.OPT_PROFILE, 1)
hd.profile_reset()
tgp.rand(eps, tgp.replace_observation_noise_cov(tgp.posterior(model, y), Rn))
prof = {k: v["total_ms"] / v["calls"] for k, v in hd.profile().items()}
print(f"{kname} d={d} T={T}: one launch, device-resident {ms1:.3f} ms ({T / ms1 * 1e3:.3e} steps/s; kernels {prof}); from host arrays {ms1h:.1f} ms; "
      f"evaluated route from host arrays {ms2:.1f} ms; max |difference| {float(np.abs(y1.cpu().numpy() - y2).max()):.2e} (scale {float(np.abs(y2).max()):.2f}); "
      f"bytes per step {8 * (d + 3)}")
