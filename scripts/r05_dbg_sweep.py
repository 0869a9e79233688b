import numpy as np
import temporalgps_jl_amd as tgp
from tests import _util as U
from tests.test_gpu_sweep import _lti_device_model, _reference, KERNELS
k, dt, s2 = KERNELS[0]
T = 2048
model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=0)
rng = np.random.default_rng(100)
missing = rng.random(T) < 0.1
lp, pm, pv = _reference(model, y, missing, 1e-18)
dm = _lti_device_model(tgp, model)
yin = np.where(missing, np.nan, y)
got, mean, var = tgp.logpdf_and_posterior_marginals(dm, yin, np.array([1e-18]))
print(dm.handle().sweep_info())
print(got, lp, np.abs(mean - pm).max(), np.abs(var - pv).max())
bad = np.nonzero(np.abs(mean - pm) > 1e-8)[0]
print(len(bad), bad[:40])
