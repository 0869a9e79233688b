import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib
from tests import _util as U
from oracle import lgssm_ref as ref
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(1)
T = 3000
model = U.random_lgssm(rng, False, d, T)
y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
hd = dm.handle()
print("RESULT auto variant", hd.lib.tgp_kernel_variant(hd.h), flush=True)
Rn = rng.random(T) * 0.1
post = ref.posterior(model, y)
pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
for variant in (1, 3):
    for chunk in (4, 11):
        hd.set_option(_lib.OPT_VARIANT, variant); hd.set_option(_lib.OPT_CHUNK, chunk)
        hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset()
        try:
            gm, gv = tgp.posterior_marginals(dm, y, Rn)
            names = sorted(hd.profile())
            print(f"RESULT variant={variant} chunk={chunk} mean err {np.max(np.abs(gm-pm)):.2e} var err {np.max(np.abs(gv-pC)):.2e} {[n for n in names if 'smooth' in n or 'posterior' in n]}", flush=True)
        except Exception as ex:
            print("RESULT variant", variant, chunk, "EXC", repr(ex)[:200], flush=True)
        hd.set_option(_lib.OPT_PROFILE, 0)
