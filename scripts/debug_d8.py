import sys
import numpy as np
sys.path.insert(0, ".")
import temporalgps_jl_amd as tgp
from oracle import lgssm_ref as ref
from tests import _util as U
from tests.test_gpu_parity import to_device_model
for d in (7, 8):
    for tv in (True, False):
        rng = np.random.default_rng(10 * d + tv)
        T = 300
        model = U.random_lgssm(rng, tv, d, T)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        y = ref.rand(model, *eps)
        lp = ref.logpdf(model, y)
        for chunk in (2, 8, 16, 64, 512):
            dm = to_device_model(tgp, model)
            dm.handle().set_option(tgp._lib.OPT_CHUNK, chunk)
            got = tgp.logpdf(dm, y)
            m, P = tgp._filter(dm, y)
            fm, fP = ref.filter_(model, y)
            bad = np.nonzero(np.abs(m - fm).max(axis=1) > 1e-7)[0]
            print(f"d={d} tv={tv} chunk={chunk}: dlogpdf={got - lp:.3e} first bad filter step={bad[:5]} n_bad={len(bad)}")
