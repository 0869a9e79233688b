"""Fixed cost of the sweep kernels: the same number of waves (1009) with chunks of 160 steps (T = 1e7) and of 320 steps (T = 2e7)."""
import sys
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P
dev = "cuda:0"
L = tgp._lib
k = P.to_kernel(("matern52",))
out = {}
for T, C in ((10_000_000, 160), (20_000_000, 320), (5_000_000, 80)):
    gen = torch.Generator(device=dev)
    gen.manual_seed(98)
    y = torch.randn((T,), dtype=torch.float64, device=dev, generator=gen)
    miss = torch.rand((T,), device=dev, generator=gen) < 0.1
    Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device=dev)
    model = P.build_lgssm(k, P.RegularSpacing(0.0, 0.1, T), 0.1)
    hd = model.handle()
    hd.set_option(L.OPT_SWEEP_CHUNK, C)
    hd.set_option(L.OPT_SWEEP_WARMUP, 8)
    hd.set_option(L.OPT_SWEEP_WARMUP_BACK, 8)
    for name, fn in (("logpdf", lambda: tgp.logpdf(model, (y, miss))), ("post", lambda: tgp.logpdf_and_posterior_marginals(model, (y, miss), Rnew))):
        fn()
        hd.set_option(L.OPT_PROFILE, 1)
        hd.profile_reset()
        for _ in range(4):
            fn()
        hd.set_option(L.OPT_PROFILE, 0)
        ms = [v["total_ms"] / v["calls"] for kk, v in hd.profile().items() if kk.startswith("k_sweep")][0]
        out[(T, name)] = ms
        print(T, C, name, round(ms, 4), hd.sweep_info()["waves"], flush=True)
    del model, y, miss
for name in ("logpdf", "post"):
    a, b, c = out[(5_000_000, name)], out[(10_000_000, name)], out[(20_000_000, name)]
    print(name, "per step (160 -> 320):", round((c - b) / 160 * 1e3, 4), "us; fixed:", round(b - 168 * (c - b) / 160, 4), "ms;  per step (80 -> 160):", round((b - a) / 80 * 1e3, 4), "us; fixed:", round(a - 88 * (b - a) / 80, 4))
