import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, _lib
SPECS = {5: ("sum", ("matern52",), ("matern32",)), 6: ("sum", ("matern52",), ("matern52",)),
         7: ("sum", ("matern52",), ("matern32",), ("matern32",)), 8: ("sum", ("matern52",), ("matern52",), ("matern32",))}
T = 10_000_000
for d in (6, 8):
    model = lti_sde.build_lgssm(lti_sde.to_kernel(SPECS[d]), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
    hd = model.handle()
    yd = torch.randn(T, dtype=torch.float64, device="cuda:0")
    hd.set_option(_lib.OPT_GROUP, 1)
    for chunk in (153, 306, 611, 1221, 2442):
        hd.set_option(_lib.OPT_CHUNK, chunk)
        for _ in range(2): tgp.logpdf(model, yd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): tgp.logpdf(model, yd)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 4 * 1e3
        hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset()
        for _ in range(2): tgp.logpdf(model, yd)
        prof = hd.profile(); hd.set_option(_lib.OPT_PROFILE, 0)
        print(f"RESULT d={d} chunk={chunk} logpdf {wall:.3f} ms | " + " ".join(f"{k.replace('k_','')}={v['total_ms']/v['calls']*1e3:.0f}" for k, v in prof.items()))
