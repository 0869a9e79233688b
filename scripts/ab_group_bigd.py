"""d = 9..16 LTI models: logpdf through the sixteen-lanes-per-chunk kernels (TGP_OPT_GROUP default) against the out-of-line
lane-per-chunk build (TGP_OPT_GROUP 0); accuracy against the sequential oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib
from tests import _util as U
from oracle import lgssm_ref as ref
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
import gc
dm = hd = None
for d in (9, 12, 14, 16):
    dm = hd = None
    gc.collect()
    torch.cuda.synchronize()
    rng = np.random.default_rng(d)
    model = U.random_lgssm(rng, False, d, T)
    y = rng.standard_normal(T)
    lp_ref = ref.logpdf(dict(model, T=2000, ), y[:2000]) if False else None
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    hd = dm.handle()
    yd = torch.as_tensor(y, device="cuda:0")
    out = {}
    for grp in (1, 0):
        hd.set_option(_lib.OPT_GROUP, grp)
        tgp.logpdf(dm, yd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out[grp] = tgp.logpdf(dm, yd)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset()
        tgp.logpdf(dm, yd)
        prof = hd.profile(); hd.set_option(_lib.OPT_PROFILE, 0)
        print(flush=True, end=""); print(f"RESULT d={d} T={T} group={grp} logpdf {wall:.2f} ms | " + " ".join(f"{k.replace('k_','')}={v['total_ms']/v['calls']*1e3:.0f}" for k, v in prof.items()))
    print(f"RESULT d={d} group vs lane-per-chunk rel diff {abs(out[1]-out[0])/abs(out[0]):.2e}")
