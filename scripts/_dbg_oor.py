import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import temporalgps_jl_amd as tgp
from tests import _util as U
from tests.test_gpu_parity import to_device_model
T, d = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
model = U.random_lgssm(rng, True, d, T, "R")
dm = to_device_model(tgp, model)
y = rng.standard_normal(T)
eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
for name, f in (("logpdf", lambda: tgp.logpdf(dm, y)), ("filter", lambda: tgp._filter(dm, y)), ("marginals", lambda: tgp.marginals(dm)), ("rand", lambda: tgp.rand(eps, dm))):
    print("running", name, flush=True)
    f()
    print("ok", name, flush=True)
