"""Pseudo-point (DTC / VFE) approximation for separable space-time kernels in state-space form, mirroring

    /root/reference/src/space_time/pseudo_point.jl:1-54    DTCSeparable, dtcify, dtc
    /root/reference/src/space_time/pseudo_point.jl:61-105  elbo, kernel_diagonals
    /root/reference/src/space_time/pseudo_point.jl:107-144 lgssm_components(::DTCSeparable, ::RectilinearGrid)
    /root/reference/src/space_time/pseudo_point.jl:187-196 build_emissions (BottleneckLGC over a LargeOutputLGC fan-out)
    /root/reference/src/space_time/pseudo_point.jl:198-235, 315-364  approx_posterior_marginals, dtc_post_emissions

The M spatial pseudo-inputs z are replicated (implicitly) at every time, the latent state is the time-kernel state of
each pseudo-point (dimension M * d_t per Separable term), the N observations of a time step see it through the
bottleneck  y = C' (H x) + noise  with  C = (K_z + 1e-12 I)^-1 K_zx.  Host code builds those few small blocks
(they are shared across time for RegularSpacing); the O(T) recursions -- logpdf, prior marginals, posterior, posterior
marginals -- run on the device through the vector-observation path (`lgssm.BottleneckLGC`).

Covered: k = sum_i s_i * Separable(k_space_i, k_time_i) on a RectilinearGrid, diagonal noise, missing observations,
state dimension sum_i M * d_t,i <= 16 and N <= 64 observations per time step (the engine's per-lane limits).
`space_time.RegularInTime` inputs -- a different number of points (at different locations) per time step,
regular_in_time.jl:8-89 -- are padded to the longest slice with the padding marked missing: per-step emission blocks, the same
device recursions. Not covered: the per-time `approx_posterior_marginals(..., t)` convenience method.
"""
import numpy as np
from scipy.linalg import block_diag

from . import lgssm as L
from . import lti_sde as S
from . import space_time as ST


class DTCSeparable(S.Kernel):
    """pseudo_point.jl:8-11."""

    def __init__(self, z, k):
        self.z, self.k = np.asarray(z, dtype=np.float64), k


def dtcify(z, k):
    """pseudo_point.jl:20-31: replace every Separable in the kernel expression by a DTCSeparable."""
    if isinstance(k, ST.Separable):
        return DTCSeparable(z, k)
    if isinstance(k, S.ScaledKernel):
        return S.ScaledKernel(k.sigma2, dtcify(z, k.kernel))
    if isinstance(k, S.KernelSum):
        return S.KernelSum(*[dtcify(z, kk) for kk in k.kernels])
    raise TypeError(f"dtcify: unsupported kernel {type(k).__name__}")


def _terms(k, scale=1.0):
    """flatten a dtcified expression into [(s_i, DTCSeparable_i)]"""
    if isinstance(k, DTCSeparable):
        return [(scale, k)]
    if isinstance(k, S.ScaledKernel):
        return _terms(k.kernel, scale * k.sigma2)
    if isinstance(k, S.KernelSum):
        return [t for kk in k.kernels for t in _terms(kk, scale)]
    raise TypeError(f"not a dtcified separable kernel: {type(k).__name__}")


def _cross(k_space, a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return ST.kappa(k_space, np.abs(a[:, None] - b[None, :]))


def _stack(blocks_per_term, join):
    n = max(b.shape[0] for b in blocks_per_term)
    ex = [b if b.shape[0] == n else np.repeat(b, n, axis=0) for b in blocks_per_term]
    return np.stack([join([e[i] for e in ex]) for i in range(n)])


def lgssm_components(k_dtc, grid, x_space=None, jitter=1e-12):
    """pseudo_point.jl:107-144 with the ScaledKernel / KernelSum rules of lti_sde.jl:344-346, 424-436.
    Returns A, a, Q, (Ct (N, Mtot), Hb (n, Mtot, D), hb (1, Mtot)), (m0, P0). `x_space` overrides the grid's spatial
    points (new prediction locations)."""
    ragged = isinstance(grid, ST.RegularInTime) and x_space is None
    xr = (grid.points() if ragged else grid.xl) if x_space is None else np.asarray(x_space, dtype=np.float64)
    As, as_, Qs, Hbs, Cs, ms, Ps = [], [], [], [], [], [], []
    for s, kd in _terms(k_dtc):
        A_t, a_t, Q_t, H_t, h_t, (m_t, P_t) = kd.k.r.lgssm_components(grid.xr)
        M = len(kd.z)
        ident = np.eye(M)
        Kz = _cross(kd.k.l, kd.z, kd.z)
        if ragged:      # one cross-covariance block per time slice (the padding slots' columns are never used: they are missing)
            Kzx = ST.kappa(kd.k.l, np.abs(kd.z[None, :, None] - xr[:, None, :]))           # (T, M, nmax)
            Cs.append(np.linalg.solve((Kz + jitter * ident)[None], Kzx))
        else:
            Cs.append(np.linalg.solve(Kz + jitter * ident, _cross(kd.k.l, kd.z, xr)))
        As.append(np.stack([np.kron(ident, Ai) for Ai in A_t]))
        as_.append(np.stack([np.tile(ai, M) for ai in a_t]))
        Qs.append(np.stack([np.kron(Kz, Qi) for Qi in Q_t]))
        Hbs.append(np.stack([np.sqrt(s) * np.kron(ident, Hi[None, :]) for Hi in np.atleast_2d(H_t)]))
        ms.append(np.tile(m_t, M))
        Ps.append(np.kron(Kz, P_t))
    A = _stack(As, lambda bs: block_diag(*bs))
    a = _stack(as_, np.concatenate)
    Q = _stack(Qs, lambda bs: block_diag(*bs))
    Hb = _stack(Hbs, lambda bs: block_diag(*bs))
    Ct = np.swapaxes(np.concatenate(Cs, axis=1), 1, 2) if ragged else np.concatenate(Cs, axis=0).T      # ragged: (T, nmax, Mtot)
    return A, a, Q, (Ct, Hb, np.zeros((1, Hb.shape[1]))), (np.concatenate(ms), block_diag(*Ps))


def kernel_diagonals(k_dtc, grid, x_space=None):
    """pseudo_point.jl:83-105: prior variances at the grid points, (T, N)."""
    T = len(grid.xr)
    n = grid.shape2[1] if x_space is None else len(np.asarray(x_space).reshape(-1))      # (stationary kernels: k(x, x) is the same at every point)
    out = np.zeros((T, n))
    for s, kd in _terms(k_dtc):
        out += s * float(S_kappa0(kd.k.r)) * ST.kappa(kd.k.l, np.zeros(n))[None, :]
    return out


def S_kappa0(k_time):
    """k_time(t, t): the stationary variance H P_inf H' of the time kernel's SDE."""
    F, H, m, P = k_time.sde_blocks()
    return float(H @ P @ H)


def _noise(grid, sigma2s):
    T, N = grid.shape2
    s = np.asarray(sigma2s, dtype=np.float64)
    if s.ndim == 0:
        return np.full((1, N), float(s))
    return grid.pad(s, 1.0) if isinstance(grid, ST.RegularInTime) else s.reshape(T, N)


def build_lgssm(k, grid, z, sigma2s, device=0):
    """build_lgssm(dtcify(z, fx)) (pseudo_point.jl:33, lti_sde.jl:71-80)."""
    k_dtc = dtcify(z, k)
    A, a, Q, (Ct, Hb, hb), (m0, P0) = lgssm_components(k_dtc, grid)
    T, N = grid.shape2
    fan_out = L.LargeOutputLGC(Ct if Ct.ndim == 3 else Ct[None], np.zeros((1, N)), _noise(grid, sigma2s))
    trans = L.GaussMarkovModel(L.Forward, A, a, Q, L.Gaussian(m0, P0))
    return L.LGSSM(trans, L.BottleneckLGC(Hb, hb, fan_out), T=T, device=device)


def _obs(grid, y):
    T, N = grid.shape2
    if isinstance(grid, ST.RegularInTime):
        return grid.pad(y, np.nan)                                # the padding slots are missing observations
    return np.asarray(y, dtype=np.float64).reshape(T, N)        # NaN == missing


def dtc(k, grid, sigma2s, y, z, device=0):
    """dtc(fx, y, z_r) = logpdf(dtcify(z_r, fx), y) (pseudo_point.jl:52-54)."""
    return L.logpdf(build_lgssm(k, grid, z, sigma2s, device), _obs(grid, y))


def elbo(k, grid, sigma2s, y, z, device=0):
    """elbo(fx, y, z_r) (pseudo_point.jl:61-81): DTC minus the trace term, which needs the diagonal of the approximate
    model's prior marginals (marginals_diag, lgssm.jl:125-127 -- `marginals` on the device) and the exact prior variances."""
    model = build_lgssm(k, grid, z, sigma2s, device)
    Y = _obs(grid, y)
    T, N = grid.shape2
    _, var = L.marginals(model)                                 # H P H' + noise, (T, N)
    var = np.asarray(var, dtype=np.float64).reshape(T, N)       # (one space point: the device hands back (T,), and (T, 1) - (T,) would broadcast to (T, T))
    Sig = np.broadcast_to(_noise(grid, sigma2s), (T, N))
    miss = np.isnan(Y)
    Sig_f = np.where(miss, 1e15, Sig)                           # fill_in_missings (missings.jl:43)
    Cf = kernel_diagonals(dtcify(z, k), grid)
    # the reference takes marg.P from the model BEFORE missings are filled in, so the noise in `var` is the original one
    tmp = np.sum((Cf - var) / Sig_f, axis=1) - miss.sum(axis=1) + N
    return L.logpdf(model, Y) - float(np.sum(tmp)) / 2.0


def build_emission_covs(k_dtc, grid, x_space, jitter=1e-9):
    """pseudo_point.jl:315-327 (+ the Scaled / Sum rules :346-363): the conditional variance of f(x*) given the
    pseudo-points, Diagonal, (T, N*)."""
    xs = np.asarray(x_space, dtype=np.float64)
    T = len(grid.xr)
    out = np.zeros((T, len(xs)))
    for s, kd in _terms(k_dtc):
        Cpu = _cross(kd.k.l, xs, kd.z)
        Ku = _cross(kd.k.l, kd.z, kd.z) + jitter * np.eye(len(kd.z))
        q = ST.kappa(kd.k.l, np.zeros(len(xs))) - np.einsum("ij,ji->i", Cpu, np.linalg.solve(Ku, Cpu.T))
        out += s * S_kappa0(kd.k.r) * q[None, :]
    return out


def approx_posterior_marginals(k, grid, sigma2s, y, z, x_r, device=0):
    """approx_posterior_marginals(dtc, fx, y, z_r, x_r) (pseudo_point.jl:198-235): DTC posterior marginals at the
    times of `grid` and the NEW spatial locations x_r. Returns (mean, var), each (T, len(x_r)) (flat order: space fastest)."""
    model = build_lgssm(k, grid, z, sigma2s, device)
    k_dtc = dtcify(z, k)
    _, _, _, (Ct, Hb, hb), _ = lgssm_components(k_dtc, grid, x_space=x_r)
    Sig = build_emission_covs(k_dtc, grid, x_r)
    if Hb.shape[0] == 1 and len(x_r) <= 4096:
        # fast path: the smoothed state straight through the new emission block H = C_new' Hb (nothing is materialised);
        # the engine offers it for the models its group-per-chunk smoother serves (state dimension 5..16)
        try:
            return L.posterior_marginals_at(model, _obs(grid, y), Ct @ Hb[0], Ct @ hb[0], Sig)
        except L._lib.Unsupported:
            pass
    post = L.posterior(model, _obs(grid, y))                    # Reverse-ordered LGSSM (lgssm.jl:193-221), device pass
    fan_out = L.LargeOutputLGC(Ct[None], np.zeros((1, len(x_r))), Sig)
    new_post = L.LGSSM(post.transitions, L.BottleneckLGC(Hb, hb, fan_out), T=model.T, device=device)
    return L.marginals(new_post)
