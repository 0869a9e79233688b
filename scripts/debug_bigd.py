import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib
from tests import _util as U
d, T, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(d)
model = U.random_lgssm(rng, False, d, T)
y = rng.standard_normal(T)
tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
hd = dm.handle()
print("RESULT handle ok", d, T, chunk, flush=True)
hd.set_option(_lib.OPT_CHUNK, chunk)
import os
hd.set_option(_lib.OPT_GROUP, int(os.environ.get("GROUP", "1")))
hd.set_option(_lib.OPT_PROFILE, 1)
lp = tgp.logpdf(dm, torch.as_tensor(y, device="cuda:0"))
print("RESULT", d, T, chunk, lp, {k: round(v["total_ms"] / v["calls"] * 1e3) for k, v in hd.profile().items()}, flush=True)
