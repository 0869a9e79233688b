"""ORACLE (test infrastructure only). Dense (O(N^3)) Gaussian-process oracle, independent of any
state-space code. It plays the role that the naive AbstractGPs GP plays in the reference's own tests
(`fx_naive` in /root/reference/test/gp/lti_sde.jl:181-200 and test/gp/posterior_lti_sde.jl:62-89):
the state-space result must equal the dense-GP result.

Third-party arithmetic restated (not under /root/reference): KernelFunctions.jl closed forms
(compat "0.9, 0.10.1", Project.toml) and AbstractGPs.jl FiniteGP logpdf / posterior (compat "0.5.17"):
    Matern12:  exp(-|tau|)
    Matern32:  (1 + sqrt3 |tau|) exp(-sqrt3 |tau|)
    Matern52:  (1 + sqrt5 |tau| + 5 tau^2 / 3) exp(-sqrt5 |tau|)
    Constant:  c
    Periodic:  exp(-0.5 sin^2(pi tau) / r^2)      (the kernel ApproxPeriodicKernel approximates)
    sigma2 * k, k o ScaleTransform(s) (tau -> s tau), sums and products.
The reference's `cosine` SDE (lti_sde.jl:239-252) has stationary covariance cos(tau); that is what the
oracle uses for ("cosine",) so that it matches the SDE the reference builds.
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve

LOG2PI = float(np.log(2.0 * np.pi))


def kappa(k, tau):
    name = k[0]
    at = np.abs(tau)
    if name == "matern12":
        return np.exp(-at)
    if name == "matern32":
        return (1 + np.sqrt(3.0) * at) * np.exp(-np.sqrt(3.0) * at)
    if name == "matern52":
        return (1 + np.sqrt(5.0) * at + 5.0 * at ** 2 / 3.0) * np.exp(-np.sqrt(5.0) * at)
    if name == "se":                       # KernelFunctions SEKernel: exp(-tau^2 / 2)
        return np.exp(-0.5 * at ** 2)
    if name == "cosine":
        return np.cos(at)
    if name == "constant":
        return np.full_like(at, float(k[1]))
    if name == "approx_periodic":
        return np.exp(-0.5 * np.sin(np.pi * at) ** 2 / k[2] ** 2)
    if name == "scaled":
        return k[1] * kappa(k[2], tau)
    if name == "stretched":
        return kappa(k[2], k[1] * tau)
    if name == "sum":
        return sum(kappa(kk, tau) for kk in k[1:])
    if name == "product":
        out = np.ones_like(at)
        for kk in k[1:]:
            out = out * kappa(kk, tau)
        return out
    raise ValueError(name)


def kernelmatrix(k, x1, x2=None):
    x2 = x1 if x2 is None else x2
    return kappa(k, x1[:, None] - x2[None, :])


def _mean(mean, x):
    if mean is None or mean[0] == "zero":
        return np.zeros(len(x))
    if mean[0] == "const":
        return np.full(len(x), float(mean[1]))
    return np.array([mean[1](v) for v in x], dtype=np.float64)


def _noise(sigma2, n):
    s = np.atleast_1d(np.asarray(sigma2, dtype=np.float64))
    return s if len(s) == n else np.full(n, s[0])


def logpdf(k, x, sigma2, y, mean=None):
    n = len(x)
    K = kernelmatrix(k, x) + np.diag(_noise(sigma2, n))
    c = cho_factor(K, lower=True)
    r = np.asarray(y) - _mean(mean, x)
    logdet = 2.0 * np.sum(np.log(np.diag(c[0])))
    return float(-(n * LOG2PI + logdet + r @ cho_solve(c, r)) / 2.0)


def marginals(k, x, sigma2, mean=None):
    n = len(x)
    return _mean(mean, x), np.diag(kernelmatrix(k, x)) + _noise(sigma2, n)


def posterior_marginals(k, x_tr, sigma2_tr, y_tr, x_pr, sigma2_pr=0.0, mean=None):
    ntr, npr = len(x_tr), len(x_pr)
    K = kernelmatrix(k, x_tr) + np.diag(_noise(sigma2_tr, ntr))
    c = cho_factor(K, lower=True)
    Ks = kernelmatrix(k, x_pr, x_tr)
    mu = _mean(mean, x_pr) + Ks @ cho_solve(c, np.asarray(y_tr) - _mean(mean, x_tr))
    var = np.diag(kernelmatrix(k, x_pr)) - np.einsum("ij,ji->i", Ks, cho_solve(c, Ks.T))
    return mu, var + _noise(sigma2_pr, npr)


def posterior_logpdf(k, x_tr, sigma2_tr, y_tr, x_pr, sigma2_pr, y_pr, mean=None):
    ntr, npr = len(x_tr), len(x_pr)
    K = kernelmatrix(k, x_tr) + np.diag(_noise(sigma2_tr, ntr))
    c = cho_factor(K, lower=True)
    Ks = kernelmatrix(k, x_pr, x_tr)
    mu = _mean(mean, x_pr) + Ks @ cho_solve(c, np.asarray(y_tr) - _mean(mean, x_tr))
    C = kernelmatrix(k, x_pr) - Ks @ cho_solve(c, Ks.T) + np.diag(_noise(sigma2_pr, npr))
    cc = cho_factor(C, lower=True)
    r = np.asarray(y_pr) - mu
    return float(-(npr * LOG2PI + 2 * np.sum(np.log(np.diag(cc[0]))) + r @ cho_solve(cc, r)) / 2.0)


def separable_kernelmatrix(k_space, k_time, r, t):
    """Separable kernel on a rectilinear grid, flat order = space fastest (rectilinear_grid.jl:5-9,31-33):
    K[(q,p),(q',p')] = k_space(r_p, r_p') k_time(t_q, t_q')  ==  kron(K_t, K_r)."""
    return np.kron(kernelmatrix(k_time, np.asarray(t, dtype=np.float64)), kernelmatrix(k_space, np.asarray(r, dtype=np.float64)))


def mvn_logpdf(K, y):
    c = cho_factor(K, lower=True)
    n = len(y)
    return float(-(n * LOG2PI + 2.0 * np.sum(np.log(np.diag(c[0]))) + y @ cho_solve(c, y)) / 2.0)


def mvn_posterior_marginals(K, noise, y, noise_new):
    """same-inputs posterior marginals of a zero-mean GP with prior covariance K and diagonal noise."""
    c = cho_factor(K + np.diag(noise), lower=True)
    mu = K @ cho_solve(c, y)
    var = np.diag(K) - np.einsum("ij,ji->i", K, cho_solve(c, K))
    return mu, var + noise_new


# ------------------------------------------------------------------ sparse (pseudo-point) approximations, dense form
# AbstractGPs.jl (compat "0.5.17", not under /root/reference) sparse_approximations.jl, restated from the published
# formulas (Titsias 2009; Quinonero-Candela & Rasmussen 2005): with u = f(z), Q_ff = K_fu K_uu^-1 K_uf,
#   DTC  approx_log_evidence = log N(y; m, Q_ff + S)
#   VFE  elbo                = DTC - tr(S^-1 (K_ff - Q_ff)) / 2
#   VFE  posterior at x*     : mean K_*u B^-1 K_uf S^-1 y,  var k_** - K_*u K_uu^-1 K_u* + K_*u B^-1 K_u*,  B = K_uu + K_uf S^-1 K_fu
# These play the role of `dtc_naive`, `elbo_naive`, `f_approx_post_naive` in test/space_time/pseudo_point.jl:92-108.
def _sep_K(terms, x1, x2):
    """x = (r (N,), t (N,)) flat space-time points; k = sum_i s_i k_space_i(r, r') k_time_i(t, t')."""
    r1, t1 = x1
    r2, t2 = x2
    out = np.zeros((len(r1), len(r2)))
    for s, ks, kt in terms:
        out += s * kappa(ks, r1[:, None] - r2[None, :]) * kappa(kt, t1[:, None] - t2[None, :])
    return out


def grid_points(r, t):
    """collect(RectilinearGrid(r, t)): space iterates fastest (rectilinear_grid.jl:31-33)."""
    r, t = np.asarray(r, dtype=np.float64), np.asarray(t, dtype=np.float64)
    return np.tile(r, len(t)), np.repeat(t, len(r))


def dtc_dense(terms, x, z, noise, y, jitter=1e-18):
    Kuu = _sep_K(terms, z, z) + jitter * np.eye(len(z[0]))
    Kuf = _sep_K(terms, z, x)
    Qff = Kuf.T @ np.linalg.solve(Kuu, Kuf)
    return mvn_logpdf(Qff + np.diag(noise), np.asarray(y, dtype=np.float64))


def elbo_dense(terms, x, z, noise, y, jitter=1e-18):
    Kuu = _sep_K(terms, z, z) + jitter * np.eye(len(z[0]))
    Kuf = _sep_K(terms, z, x)
    Qff_diag = np.einsum("ij,ij->j", Kuf, np.linalg.solve(Kuu, Kuf))
    kff = sum(s * kappa(ks, np.zeros(1))[0] * kappa(kt, np.zeros(1))[0] for s, ks, kt in terms)
    return dtc_dense(terms, x, z, noise, y, jitter) - 0.5 * np.sum((kff - Qff_diag) / noise)


def vfe_posterior_marginals(terms, x, z, noise, y, xs, jitter=1e-18):
    Kuu = _sep_K(terms, z, z) + jitter * np.eye(len(z[0]))
    Kuf = _sep_K(terms, z, x)
    Ksu = _sep_K(terms, xs, z)
    B = Kuu + (Kuf / noise) @ Kuf.T
    mean = Ksu @ np.linalg.solve(B, Kuf @ (np.asarray(y, dtype=np.float64) / noise))
    kss = sum(s * kappa(ks, np.zeros(1))[0] * kappa(kt, np.zeros(1))[0] for s, ks, kt in terms)
    var = kss - np.einsum("ij,ji->i", Ksu, np.linalg.solve(Kuu, Ksu.T)) + np.einsum("ij,ji->i", Ksu, np.linalg.solve(B, Ksu.T))
    return mean, var
