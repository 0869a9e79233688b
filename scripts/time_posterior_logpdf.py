"""logpdf(replace_observation_noise_cov(posterior(model, y), R_new), y_new) at T steps, device-resident series: the pair-statistic route
(tgp_pair_statistic + two prior logpdf calls) against the evaluated posterior (tgp_posterior, re-bound as a Reverse model, filtered)."""
import sys
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as S

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kernels = {1: S.Matern12Kernel(), 2: S.Matern32Kernel(), 3: S.Matern52Kernel(), 5: S.Matern32Kernel() + S.Matern52Kernel().stretch(0.7)}
dev = torch.device("cuda:0")
for d, k in kernels.items():
    fx = S.to_sde(S.GP(k), S.HIPStorage())(S.RegularSpacing(0.0, 0.1, T), 0.1)          # the bench's models: unit kernels, dt = 0.1, noise 0.1
    model = fx.build_lgssm()
    g = torch.Generator(device=dev).manual_seed(d)
    y = tgp.rand((torch.randn((T, d), dtype=torch.float64, device=dev, generator=g), torch.randn(T, dtype=torch.float64, device=dev, generator=g),
                  np.random.default_rng(d).standard_normal(d)), model)
    y_new = y + 0.3 * torch.randn(T, dtype=torch.float64, device=dev, generator=g)
    R_new = np.array([0.05])

    def pair():
        return tgp.logpdf(tgp.replace_observation_noise_cov(tgp.posterior(model, y), R_new), y_new)

    def evaluated():
        return tgp.logpdf(tgp.replace_observation_noise_cov(tgp.posterior(model, y), R_new).materialise(), y_new)

    def best(fn, n):
        out, ts = None, []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return out, min(ts) * 1e3

    _, tp = best(lambda: tgp.logpdf(model, y), 6)
    a, ta = best(pair, 6)
    try:
        b, tb = best(evaluated, 2) if T * (2 * d * d + d) * 8 < 40e9 else (float("nan"), float("nan"))
    except Exception as ex:      # noqa: BLE001
        print("evaluated route:", type(ex).__name__, ex)
        b, tb = float("nan"), float("nan")
    print(f"d = {d}, T = {T}: posterior logpdf {ta:.3f} ms through the pair statistic, {tb:.2f} ms with the posterior evaluated (the prior's logpdf alone {tp:.3f} ms); {a:.12g} / {b:.12g}", flush=True)
