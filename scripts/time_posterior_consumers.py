"""logpdf / rand of posterior(fx, y)(fx.x, s) (posterior_lti_sde.jl:48-78) from host arrays at the training inputs: the mirror's same-inputs routes
(pair statistic -> two prior logpdf calls; tgp_posterior_rand) against the joined 2T-step route the reference's call chain spells out."""
import sys
import time

import numpy as np

import temporalgps_jl_amd  # noqa: F401
from temporalgps_jl_amd import lti_sde as S

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
x = S.RegularSpacing(0.0, 0.01, T)
fx = S.to_sde(S.GP(1.5 * S.Matern52Kernel().stretch(1 / 2.3)), S.HIPStorage())(x, 0.5)
y = S.rand(rng, fx)
ys = y + 0.3 * rng.standard_normal(T)
fp = S.posterior(fx, y)(x, 0.2)


def best(fn, n=3):
    out, ts = None, []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return out, min(ts) * 1e3


a, ta = best(lambda: S.logpdf(fp, ys))
b, tb = best(lambda: fp._logpdf_merged(ys), 2)
print(f"T = {T}: posterior logpdf at the training inputs {ta:.1f} ms (joined 2T-step route {tb:.1f} ms), values {a:.9g} / {b:.9g}, rel {abs(a - b) / abs(b):.1e}")
_, tc = best(lambda: S.rand(np.random.default_rng(1), fp))
_, td = best(lambda: fp._rand_merged(np.random.default_rng(1)), 2)
_, te = best(lambda: np.random.default_rng(1).standard_normal((T, 5)))
print(f"          posterior rand at the training inputs {tc:.1f} ms (joined route {td:.1f} ms); drawing its {5 * T} normals alone {te:.1f} ms")
