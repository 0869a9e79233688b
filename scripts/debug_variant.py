import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib, lti_sde
from tests import _util as U
from oracle import lgssm_ref as ref
d = int(sys.argv[1])
rng = np.random.default_rng(0)
T = 3000
for tv in (False, True):
    model = U.random_lgssm(rng, tv, d, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    ym = y.copy(); ym[::17] = np.nan
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    hd = dm.handle()
    print("RESULT d", d, "tv", tv, "auto variant", hd.lib.tgp_kernel_variant(hd.h))
    ops = {
        "logpdf": lambda: np.array([tgp.logpdf(dm, y)]),
        "logpdf_missing": lambda: np.array([tgp.logpdf(dm, ym)]),
        "filter": lambda: np.concatenate([x.reshape(-1) for x in tgp._filter(dm, y)]),
        "posterior": lambda: (lambda p: np.concatenate([p.transitions.As.reshape(-1), p.transitions.as_.reshape(-1) if hasattr(p.transitions, "as_") else np.zeros(1), p.transitions.Qs.reshape(-1)]))(tgp.posterior(dm, y)),
        "post_marg": lambda: np.concatenate(tgp.posterior_marginals(dm, y, np.full(T, 0.05))),
        "post_marg_sharedR": lambda: np.concatenate(tgp.posterior_marginals(dm, ym, np.array([0.05]))),
        "marginals": lambda: np.concatenate(tgp.marginals(dm)),
        "rand": lambda: tgp.rand(eps, dm),
    }
    for chunk in (4, 11):
        for name, fn in ops.items():
            res = {}
            for v in (1, 2):
                hd.set_option(_lib.OPT_VARIANT, v); hd.set_option(_lib.OPT_CHUNK, chunk)
                try:
                    res[v] = np.asarray(fn())
                except Exception as ex:
                    res[v] = repr(ex)[:80]
            if isinstance(res[1], str) or isinstance(res[2], str):
                print("RESULT   chunk", chunk, name, "v1:", res[1] if isinstance(res[1], str) else "ok", "| v2:", res[2] if isinstance(res[2], str) else "ok")
            else:
                print("RESULT   chunk", chunk, name, "max abs diff %.3e" % float(np.max(np.abs(res[1] - res[2]))))
