"""Development: wall time per combined call and the hipEvent average of the kernels of one workload (the environment selects what is
measured: TGP_MODAL_ABLATE, TGP_MODAL_GEOMETRY, ...).  usage: r04_time_kernel.py workload [T] [logpdf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import temporalgps_jl_amd as tgp

name = sys.argv[1]
T = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
only_logpdf = len(sys.argv) > 3 and sys.argv[3] == "logpdf"
model = bench.build_model(tgp, name, T, "lti", 0)
hd = model.handle()
gen = torch.Generator(device="cuda:0")
gen.manual_seed(99)
y = torch.randn((T,), dtype=torch.float64, device="cuda:0", generator=gen)
Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
step = (lambda: tgp.logpdf(model, y)) if only_logpdf else (lambda: tgp.logpdf_and_posterior_marginals(model, y, Rn))
for _ in range(5):
    step()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 40)
hd.set_option(tgp._lib.OPT_PROFILE, 1)
hd.profile_reset()
for _ in range(40):
    step()
torch.cuda.synchronize()
hd.set_option(tgp._lib.OPT_PROFILE, 0)
prof = {k: round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in hd.profile().items()}
print(f"{name} T={T} {'logpdf' if only_logpdf else 'combined'}: {best * 1e3:.4f} ms/call  kernels(us) {prof}", flush=True)
