"""Posterior marginals of d = 9..16 LTI models: group-per-chunk path (default) against the out-of-line lane-per-chunk kernels."""
import sys, time, gc
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib
from tests import _util as U
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dm = hd = None
for d in (9, 14, 16):
    dm = hd = None; gc.collect(); torch.cuda.synchronize()
    rng = np.random.default_rng(d)
    model = U.random_lgssm(rng, False, d, T)
    yd = torch.as_tensor(rng.standard_normal(T), device="cuda:0")
    Rn = torch.full((1,), 0.05, dtype=torch.float64, device="cuda:0")
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    hd = dm.handle()
    out = {}
    for grp in (1, 0):
        hd.set_option(_lib.OPT_GROUP, grp)
        tgp.posterior_marginals(dm, yd, Rn)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out[grp] = tgp.posterior_marginals(dm, yd, Rn)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        print(f"RESULT d={d} T={T} group={grp} posterior marginals {wall:.1f} ms", flush=True)
    print(f"RESULT d={d} max abs diff mean {float((out[1][0]-out[0][0]).abs().max()):.2e} var {float((out[1][1]-out[0][1]).abs().max()):.2e}", flush=True)
