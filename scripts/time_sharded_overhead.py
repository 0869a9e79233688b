"""Host-side overhead of the time-sharded path (exchange + extra host round trips) with ONE rank over RCCL, against the
direct single-GPU calls on the same series."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch, torch.distributed as dist
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, parallel
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29871", rank=0, world_size=1, device_id=torch.device("cuda:0"))
T = 10_000_000
model = lti_sde.build_lgssm(lti_sde.Matern52Kernel(), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
y = torch.randn(T, dtype=torch.float64, device="cuda:0")
Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
sh = parallel.ShardedLGSSM(model, 1, 0, engine=parallel.HIPEngine(model))
def t(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e3
print("RESULT direct  logpdf %.3f ms  posterior_marginals %.3f ms" % (t(lambda: tgp.logpdf(model, y)), t(lambda: tgp.posterior_marginals(model, y, Rn))))
print("RESULT sharded logpdf %.3f ms  posterior_marginals %.3f ms" % (t(lambda: sh.logpdf(y)), t(lambda: sh.posterior_marginals(y, Rn))))
dist.destroy_process_group()
