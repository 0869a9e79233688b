"""A/B on ONE box: fused vs stand-alone level-0 scan (TGP_OPT_FUSE_SCAN), per-kernel hipEvent profile, T = 1e7, d = 3."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
model = lti_sde.build_lgssm(lti_sde.Matern52Kernel(), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
if len(sys.argv) > 2 and sys.argv[2] == "randn":
    y = torch.randn(T, dtype=torch.float64, device="cuda:0")
else:
    y = tgp.rand((torch.randn((T, 3), dtype=torch.float64, device="cuda:0"), torch.randn(T, dtype=torch.float64, device="cuda:0"),
                  np.random.default_rng(0).standard_normal(3)), model)
Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
hd = model.handle()
def run(n=10):
    for _ in range(3):
        tgp.logpdf(model, y); tgp.posterior_marginals(model, y, Rn)
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n):
        tgp.logpdf(model, y); tgp.posterior_marginals(model, y, Rn)
    torch.cuda.synchronize(); wall = (time.perf_counter() - a) / n * 1e3
    hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset()
    for _ in range(n):
        tgp.logpdf(model, y); tgp.posterior_marginals(model, y, Rn)
    prof = hd.profile(); hd.set_option(_lib.OPT_PROFILE, 0)
    return wall, prof
for rep in range(2):
    for fuse in (1, 0):
        hd.set_option(_lib.OPT_FUSE_SCAN, fuse)
        wall, prof = run()
        tot = sum(v["total_ms"] for v in prof.values()) / 10
        print(f"RESULT fuse={fuse} wall {wall:.3f} ms  kernels {tot:.3f} ms | " + " ".join(f"{k.replace('k_','')}={v['total_ms']/v['calls']*1e3:.0f}" for k, v in prof.items()))
