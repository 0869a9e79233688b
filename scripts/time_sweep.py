"""Times the sweep engine (TGP_OPT_SWEEP) against the general engine on the three predict-path workloads of the round-4 verdict, T = 1e7, d = 3
(cfg2's kernel): LTI + 10 % missing, LTI + per-step noise, irregular spacing.  One combined call (logpdf + posterior marginals) per step."""
import sys
import time

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kname = sys.argv[2] if len(sys.argv) > 2 else "matern52"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
k = P.to_kernel((kname,))
dt, s2 = 0.1, 0.1
rng = np.random.default_rng(3)
gen = torch.Generator(device=dev)
gen.manual_seed(98)
y = torch.randn((T,), dtype=torch.float64, device=dev, generator=gen)
Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device=dev)


def timed(model, yin, label):
    hd = model.handle()
    for sweep in (1, 0):
        hd.set_option(tgp._lib.OPT_SWEEP, sweep)
        for _ in range(2):
            lp, m, v = tgp.logpdf_and_posterior_marginals(model, yin, Rnew)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tgp.logpdf_and_posterior_marginals(model, yin, Rnew)
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            tgp.logpdf(model, yin)
        torch.cuda.synchronize()
        dtl = (time.perf_counter() - t0) / steps
        info = hd.sweep_info()
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        hd.profile_reset()
        tgp.logpdf_and_posterior_marginals(model, yin, Rnew)
        tgp.logpdf(model, yin)
        hd.set_option(tgp._lib.OPT_PROFILE, 0)
        prof = {kk: round(vv["total_ms"] / vv["calls"], 4) for kk, vv in hd.profile().items()}
        print(f"{label:24s} sweep={sweep} combined {dtc * 1e3:8.3f} ms  logpdf {dtl * 1e3:8.3f} ms  lml {float(lp):.6f}  info {info}  kernels {prof}", flush=True)
        if sweep == 1:
            keep = (float(lp), m.clone(), v.clone())
        else:
            print(f"{'':24s} vs general: lml rel {abs(keep[0] - float(lp)) / abs(float(lp)):.2e}  mean {float((keep[1] - m).abs().max()):.2e}  var {float((keep[2] - v).abs().max()):.2e}", flush=True)


# 1. LTI + 10 % missing
model = P.build_lgssm(k, P.RegularSpacing(0.0, dt, T), s2)
miss = torch.rand((T,), device=dev, generator=gen) < 0.1
timed(model, (y, miss), "lti_missing_10pct")
del model
# 2. LTI + per-step noise
S = s2 * (0.5 + rng.random(T))
model = P.build_lgssm(k, P.RegularSpacing(0.0, dt, T), S)
timed(model, y, "lti_per_step_noise")
del model
# 3. irregular spacing
t = np.cumsum(rng.uniform(0.5 * dt, 1.5 * dt, T))
model = P.build_lgssm(k, t, s2, device_components=True)
timed(model, y, "irregular_spacing")
timed(model, (y, miss), "irregular + missing")
