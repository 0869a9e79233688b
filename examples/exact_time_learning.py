"""Hyper-parameter learning + exact inference on a long regularly-sampled series: the counterpart of
/root/reference/examples/exact_time_learning.jl (Optim.BFGS + Mooncake there; scipy L-BFGS-B + the device's forward-mode
gradient `logpdf_and_gradient` here). T = 1e6 points; every objective evaluation is one logpdf + 4 tangent scans on the GPU.

    python examples/exact_time_learning.py [T]
"""
import sys
import time

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, ".")
import temporalgps_jl_amd as tgp  # noqa: E402,F401
from temporalgps_jl_amd import lti_sde as S  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = S.RegularSpacing(0.0, 1e-4, T)


def build_gp(p):
    """params.var_kernel * Matern52Kernel() ∘ ScaleTransform(params.λ), constant mean (exact_time_learning.jl:31-34)"""
    return S.to_sde(S.GP(float(p["mean"]), float(p["var_kernel"]) * S.Matern52Kernel().stretch(float(p["lam"]))))


true = dict(mean=3.0, var_kernel=0.6, lam=10.0, var_noise=2.0)
rng = np.random.default_rng(0)
y = S.rand(rng, build_gp(true)(x, true["var_noise"]))

names = ["mean", "var_kernel", "lam", "var_noise"]
grad_name = {"mean": "mean.c", "var_kernel": "kernel.sigma2", "lam": "kernel.kernel.s", "var_noise": "noise"}


def unpack(theta):      # positive parameters live on the log scale (ParameterHandling.positive)
    return dict(mean=theta[0], var_kernel=np.exp(theta[1]), lam=np.exp(theta[2]), var_noise=np.exp(theta[3]))


def objective(theta):
    p = unpack(theta)
    lp, g = S.logpdf_and_gradient(build_gp(p)(x, p["var_noise"]), y)
    dtheta = np.array([g[grad_name["mean"]]] + [g[grad_name[n]] * p[n] for n in names[1:]])    # chain rule through exp
    return -lp / T, -dtheta / T


theta0 = np.array([2.0, np.log(1.5), np.log(4.0), np.log(0.7)])
t0 = time.perf_counter()
res = minimize(objective, theta0, jac=True, method="L-BFGS-B", options=dict(maxiter=60))
t1 = time.perf_counter()
final = unpack(res.x)
print(f"optimiser: {res.nit} iterations, {res.nfev} objective + gradient evaluations in {t1 - t0:.2f} s "
      f"({(t1 - t0) / res.nfev * 1e3:.1f} ms each at T = {T})")
for n in names:
    print(f"  {n:11s} true {true[n]:7.3f}   learned {final[n]:7.3f}")

# exact posterior marginals at 1.2 T points (exact_time_learning.jl:69-79): prediction beyond the data
f_post = S.posterior(build_gp(final)(x, final["var_noise"]), y)
x_pr = S.RegularSpacing(0.0, 1e-4, int(1.2 * T))
mean, std = S.marginals(f_post(x_pr, 1e-18))
mean, std = np.asarray(mean.cpu() if hasattr(mean, "cpu") else mean), np.asarray(std.cpu() if hasattr(std, "cpu") else std)
print(f"posterior at {len(mean)} points: mean in [{mean.min():.3f}, {mean.max():.3f}], std in [{std.min():.4f}, {std.max():.4f}]")
