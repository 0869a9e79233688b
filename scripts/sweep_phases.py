"""Per-pass cost of the sweep engine's kernel: forced geometries (the results of a too-short warm-up are discarded by the library, the kernel's time
is what is read here).  LTI + 10 % missing, T = 1e7, d = 3 unless told otherwise."""
import sys

import numpy as np
import torch

import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "lti"
dev = "cuda:0"
k = P.to_kernel(("matern52",))
gen = torch.Generator(device=dev)
gen.manual_seed(98)
y = torch.randn((T,), dtype=torch.float64, device=dev, generator=gen)
miss = torch.rand((T,), device=dev, generator=gen) < 0.1
Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device=dev)
if mode == "lti":
    model = P.build_lgssm(k, P.RegularSpacing(0.0, 0.1, T), 0.1)
    yin = (y, miss)
else:
    t = np.cumsum(np.random.default_rng(3).uniform(0.05, 0.15, T))
    model = P.build_lgssm(k, t, 0.1, device_components=True)
    yin = y
hd = model.handle()
L = tgp._lib


def kernel_ms(fn, n=4):
    fn()
    hd.set_option(L.OPT_PROFILE, 1)
    hd.profile_reset()
    for _ in range(n):
        fn()
    hd.set_option(L.OPT_PROFILE, 0)
    return {kk: round(v["total_ms"] / v["calls"], 4) for kk, v in hd.profile().items() if kk.startswith("k_sweep")}


for C, W, Wb in [(160, 8, 8), (160, 136, 8), (160, 8, 136), (160, 136, 136), (160, 104, 96), (320, 8, 8), (80, 8, 8)]:
    hd.set_option(L.OPT_SWEEP_CHUNK, C)
    hd.set_option(L.OPT_SWEEP_WARMUP, W)
    hd.set_option(L.OPT_SWEEP_WARMUP_BACK, Wb)
    lp = kernel_ms(lambda: tgp.logpdf(model, yin))
    pm = kernel_ms(lambda: tgp.logpdf_and_posterior_marginals(model, yin, Rnew))
    print(f"C {C:4d} W {W:4d} Wb {Wb:4d}: {lp} {pm}  info {hd.sweep_info()['status']}", flush=True)
