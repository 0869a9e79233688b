import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
base = sys.argv[2] if len(sys.argv) > 2 else "matern52"
k = P.ScaledKernel(1.0, P.StretchedKernel(1 / 2.3 if False else 1.0, P.to_kernel((base,))))
fx = P.to_sde(P.GP(k))(P.RegularSpacing(0.0, 0.1, T), 0.1)
model = fx.build_lgssm()
d = model.dim
gen = torch.Generator(device="cuda:0"); gen.manual_seed(1)
y = tgp.rand((torch.randn((T, d), dtype=torch.float64, device="cuda:0", generator=gen),
              torch.randn((T,), dtype=torch.float64, device="cuda:0", generator=gen), np.zeros(d)), model)
hd = model.handle()
hd.set_option(tgp._lib.OPT_TIMING, 1)
for _ in range(2):
    lp, g = P.logpdf_and_gradient(fx, y)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); lp, g = P.logpdf_and_gradient(fx, y); ts.append(time.perf_counter() - t0)
hd.set_option(tgp._lib.OPT_PROFILE, 1); hd.profile_reset()
lp, g = P.logpdf_and_gradient(fx, y)
prof = {k_: round(v["total_ms"] / v["calls"] * 1e3, 1) for k_, v in hd.profile().items()}
hd.set_option(tgp._lib.OPT_PROFILE, 0)
t0 = time.perf_counter(); lp2 = tgp.logpdf(model, y); tl = time.perf_counter() - t0
print(json.dumps(dict(T=T, d=d, n_params=len(g), logpdf=lp, grad=g, ms_logpdf_and_grad=min(ts) * 1e3, steps_per_s=T / min(ts), ms_logpdf=tl * 1e3,
                      kernel_ms=hd.last_timing(), kernels_us=prof)))
