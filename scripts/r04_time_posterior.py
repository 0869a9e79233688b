"""Development: tgp_posterior (the evaluated reverse-time model) of an LTI model, device-resident -- the filter's one-launch kernel with the
posterior's outputs (+ the head on the host) against the general engine (TGP_OPT_STEADY = 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import temporalgps_jl_amd as tgp

T = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
name = sys.argv[1] if len(sys.argv) > 1 else "matern52_d3"
for opt in (3, 2):
    model = bench.build_model(tgp, name, T, "lti", 0)
    model.handle_options[tgp._lib.OPT_STEADY] = opt
    hd = model.handle()
    d = model.dim
    y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
    for _ in range(3):
        post = tgp.posterior(model, y).materialise()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 10
    for _ in range(N):
        post = tgp.posterior(model, y).materialise()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    for _ in range(3):
        tgp.posterior(model, y).materialise()
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    prof = {k: round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in hd.profile().items()}
    print(f"{name} T={T} option {opt}: {dt * 1e3:.4f} ms per posterior ({T / dt:.3e} steps/s; {8 * (1 + d + 2 * d * d) * T / dt / 1e12:.2f} TB/s of y + outputs)  kernels(us) {prof}", flush=True)
    del model, post
