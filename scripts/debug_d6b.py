import sys
import numpy as np
sys.path.insert(0, ".")
np.set_printoptions(precision=5, linewidth=200)
import temporalgps_jl_amd as tgp
from oracle import lgssm_ref as ref
from tests import _util as U
from tests.test_gpu_parity import to_device_model
from tests.test_golden import load_case
model, g = load_case("random_tv_d6")
y = g("y")
fm, fP = ref.filter_(model, y)
dm = to_device_model(tgp, model)
dm.handle().set_option(tgp._lib.OPT_CHUNK, 33)
m, P = tgp._filter(dm, y)
print("ref m0", fm[0]); print("dev m0", m[0])
print("ref P0 diag", np.diag(fP[0])); print("dev P0 diag", np.diag(P[0]))
mp, Pp = ref.predict(model["x0m"], model["x0P"], model["A"][0], model["a"][0], model["Q"][0])
print("ref predicted m", mp)
# hypotheses
for name, mm, PP in [("x0=0,I", np.zeros(6), np.eye(6)), ("x0 given", model["x0m"], model["x0P"])]:
    a, b = ref.predict(mm, PP, model["A"][0], model["a"][0], model["Q"][0])
    a, b, _ = ref.posterior_and_lml_scalar(a, b, model["H"][0], model["h"][0], model["R"][0], y[0])
    print(name, "->", a)
post = tgp.posterior(dm, y)
print("xf dev", post.x0.m); print("xf ref", fm[-1])
# the same model with T truncated to 20 and extended by repetition
for T2 in (8, 20, 32, 33):
    mod2 = dict(model, T=T2, **{k: model[k][:T2] for k in ("A", "a", "Q", "H", "h", "R")})
    d2 = to_device_model(tgp, mod2)
    lp = ref.logpdf(mod2, y[:T2])
    print("T", T2, "dlogpdf", tgp.logpdf(d2, y[:T2]) - lp)
