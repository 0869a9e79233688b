"""Separable space-time GPs on a rectilinear grid, mirroring

    /root/reference/src/space_time/separable_kernel.jl:9-41    Separable(k_space, k_time)
    /root/reference/src/space_time/rectilinear_grid.jl:11-97   RectilinearGrid(xl, xr): space iterates fastest
    /root/reference/src/space_time/to_gauss_markov.jl:1-20     lgssm_components(::Separable, ::SpaceTimeGrid, storage)

Two ways onto the device:

* `build_lgssm(k, grid, sigma2s)`: the reference's own construction, literally -- one LGSSM with state dimension
  Nr * d_t and Nr observations per time step (SmallOutputLGC, diagonal noise). The per-lane engine covers
  Nr * d_t <= 8.
* `DecoupledSpaceTime`: an exact algebraic shortcut the reference does NOT take, for the common
  case "full grid, noise variance equal across space at every time, zero mean, no missing data": with
  K_r = V diag(lam) V', the rotated observations Y V are Nr INDEPENDENT scalar-output time series whose kernels
  are lam_i * k_time, so logpdf and the posterior marginals of the Nr * d_t-dimensional model come from Nr runs
  of the d_t-dimensional scalar engine, concatenated into ONE device series with a state reset between
  components. logpdf is identical (orthogonal change of variables); marginals are rotated back.
  At Nr = 256, T = 1e5 (BASELINE config 5) this is 2.56e7 scalar Kalman steps instead of 1e5 dense d = 768 steps.
"""
import numpy as np

from . import lgssm as L
from . import lti_sde as S


class SEKernel(S.Kernel):
    """exp(-tau^2 / 2): spatial use only (no finite-dimensional SDE)."""

    def kappa(self, tau):
        return np.exp(-0.5 * np.asarray(tau) ** 2)


def kappa(k, tau):
    """Stationary kernel value at distance tau (KernelFunctions.jl closed forms), for the SPATIAL factor."""
    at = np.abs(np.asarray(tau, dtype=np.float64))
    if isinstance(k, SEKernel):
        return k.kappa(at)
    if isinstance(k, S.Matern12Kernel):
        return np.exp(-at)
    if isinstance(k, S.Matern32Kernel):
        return (1 + np.sqrt(3.0) * at) * np.exp(-np.sqrt(3.0) * at)
    if isinstance(k, S.Matern52Kernel):
        return (1 + np.sqrt(5.0) * at + 5.0 * at ** 2 / 3.0) * np.exp(-np.sqrt(5.0) * at)
    if isinstance(k, S.ConstantKernel):
        return np.full_like(at, k.c)
    if isinstance(k, S.ScaledKernel):
        return k.sigma2 * kappa(k.kernel, at)
    if isinstance(k, S.StretchedKernel):
        return kappa(k.kernel, k.s * at)
    if isinstance(k, S.KernelSum):
        return sum(kappa(kk, at) for kk in k.kernels)
    if isinstance(k, S.KernelProduct):
        out = np.ones_like(at)
        for kk in k.kernels:
            out = out * kappa(kk, at)
        return out
    raise TypeError(f"no closed form for {type(k).__name__}")


def kernelmatrix(k, x):
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        dist = np.abs(x[:, None] - x[None, :])
    else:
        dist = np.linalg.norm(x[:, None, :] - x[None, :, :], axis=-1)
    return kappa(k, dist)


class Separable(S.Kernel):
    def __init__(self, l, r):
        self.l, self.r = l, r          # l: space, r: time (separable_kernel.jl:9-12)


class RectilinearGrid:
    """xl: spatial points, xr: times; flat index n -> (xl[n % Nl], xr[n // Nl])."""

    def __init__(self, xl, xr):
        self.xl, self.xr = np.asarray(xl, dtype=np.float64), xr

    def __len__(self):
        return len(self.xl) * len(self.xr)

    @property
    def shape2(self):
        return len(self.xr), len(self.xl)      # (T, Nr): observations reshape to this, row-major


class RegularInTime:
    """regular_in_time.jl:8-89: several observations at each of a collection of time slices, a DIFFERENT number (and different
    locations) per slice allowed. ts: the times (RegularSpacing or array), vs: one array of spatial points per time.
    Flat order of observations / noise variances: slice after slice (`collect`, regular_in_time.jl:22-26).
    The device path has a fixed number of observations per step: slices are padded to the longest one, the padding marked missing
    (`pad` / `pad_mask`), which is exactly marginalising it out (missings.jl:8-33)."""

    def __init__(self, ts, vs):
        self.xr = ts
        self.vs = [np.asarray(v, dtype=np.float64).reshape(-1) for v in vs]
        if len(self.vs) != len(ts):
            raise ValueError("RegularInTime: one array of spatial points per time")
        self.lengths = np.array([len(v) for v in self.vs])
        self.nmax = int(self.lengths.max())

    def __len__(self):
        return int(self.lengths.sum())

    @property
    def shape2(self):
        return len(self.vs), self.nmax

    @property
    def pad_mask(self):
        """(T, nmax) True where a slot holds no observation"""
        return np.arange(self.nmax)[None, :] >= self.lengths[:, None]

    def pad(self, flat, fill):
        """observations_to_time_form / noise_var_to_time_form (regular_in_time.jl:53-63), then padded to (T, nmax)"""
        flat = np.asarray(flat, dtype=np.float64).reshape(-1)
        if flat.size != len(self):
            raise ValueError(f"expected {len(self)} values (one per observation), got {flat.size}")
        out = np.full(self.shape2, fill, dtype=np.float64)
        out[~self.pad_mask] = flat
        return out

    def unpad(self, padded):
        """destructure (regular_in_time.jl:65): back to the flat order"""
        return np.asarray(padded)[~self.pad_mask]

    def points(self, fill=0.0):
        """(T, nmax) spatial points, padding slots at `fill`"""
        out = np.full(self.shape2, fill, dtype=np.float64)
        out[~self.pad_mask] = np.concatenate(self.vs)
        return out


def lgssm_components(k, grid):
    """to_gauss_markov.jl:1-20."""
    Kr = kernelmatrix(k.l, grid.xl)
    A_t, a_t, Q_t, H_t, h_t, (m_t, P_t) = k.r.lgssm_components(grid.xr)
    Nr = len(grid.xl)
    ident = np.eye(Nr)
    A = np.stack([np.kron(ident, Ai) for Ai in A_t])
    a = np.stack([np.tile(ai, Nr) for ai in a_t])
    Q = np.stack([np.kron(Kr + 1e-12 * ident, Qi) for Qi in Q_t])
    H = np.stack([np.kron(ident, Hi[None, :]) for Hi in H_t])
    h = np.stack([np.full(Nr, hi) for hi in np.atleast_1d(h_t)])
    return A, a, Q, H, h, (np.tile(m_t, Nr), np.kron(Kr, P_t))


def build_lgssm(k, grid, sigma2s, device=0):
    """The reference's literal construction; sigma2s: scalar or (T, Nr) / flat (T*Nr,) noise variances."""
    A, a, Q, H, h, (m0, P0) = lgssm_components(k, grid)
    T, Nr = grid.shape2
    s = np.asarray(sigma2s, dtype=np.float64)
    R = np.full((1, Nr), float(s)) if s.ndim == 0 else s.reshape(T, Nr)
    trans = L.GaussMarkovModel(L.Forward, A, a, Q, L.Gaussian(m0, P0))
    return L.LGSSM(trans, L.SmallOutputLGC(H, h, R), T=T, device=device)


class DecoupledSpaceTime:
    """Eigen-decoupled exact inference for Separable kernels on a full grid (see the module docstring)."""

    def __init__(self, k, grid, sigma2, device=0):
        T, Nr = grid.shape2
        s = np.asarray(sigma2, dtype=np.float64)
        if s.ndim == 0:
            s_t = np.full(T, float(s))
        else:
            s2 = s.reshape(T, Nr)
            if not np.allclose(s2, s2[:, :1]):
                raise ValueError("the decoupled path needs a noise variance that is equal across space at every time")
            s_t = s2[:, 0].copy()
        self.T, self.Nr, self.grid = T, Nr, grid
        Kr = kernelmatrix(k.l, grid.xl)
        lam, V = np.linalg.eigh(Kr)
        self.lam, self.V = np.clip(lam, 0.0, None), V
        A_t, a_t, Q_t, H_t, h_t, (m0, P0) = k.r.lgssm_components(grid.xr)
        d = len(m0)
        rep = lambda z: z if z.shape[0] == T else np.repeat(z, T, axis=0)
        A_t, a_t, Q_t, H_t = rep(A_t), rep(a_t), rep(Q_t), rep(H_t)
        # one long series: component i occupies steps [i*T, (i+1)*T); its first step restarts from the prior:
        #   x_1 = 0 * x_prev + (A_1 m0 + a_1) + N(0, A_1 P0 A_1' + Q_1)
        A = np.tile(A_t, (Nr, 1, 1))
        a = np.tile(a_t, (Nr, 1))
        Q = np.tile(Q_t, (Nr, 1, 1))
        first = np.arange(Nr) * T
        a[first] = A_t[0] @ m0 + a_t[0]
        Q[first] = A_t[0] @ P0 @ A_t[0].T + Q_t[0]
        A[first] = 0.0
        H = (np.sqrt(self.lam)[:, None, None] * H_t[None]).reshape(Nr * T, d)     # ScaledKernel scales H (lti_sde.jl:334-342)
        R = np.tile(s_t, Nr)
        x0 = L.Gaussian(np.zeros(d), np.eye(d))                                   # irrelevant: step 1 resets
        self.model = L.LGSSM(L.GaussMarkovModel(L.Forward, A, a, Q, x0), L.ScalarOutputLGC(H, np.zeros(1), R), T=Nr * T,
                             device=device)

    def _rotate(self, y):
        Y = np.asarray(y, dtype=np.float64).reshape(self.T, self.Nr)
        if np.isnan(Y).any():
            raise ValueError("the decoupled path does not support missing observations")
        return np.ascontiguousarray((Y @ self.V).T).reshape(-1)                    # component-major series

    def logpdf(self, y):
        return L.logpdf(self.model, self._rotate(y))

    def posterior_marginals(self, y, sigma2_new=1e-18):
        """(mean, var) of marginals(posterior(fx, y)(x, sigma2_new)) on the same grid, shape (T, Nr)."""
        m, v = L.posterior_marginals(self.model, self._rotate(y), np.zeros(1))
        M = m.reshape(self.Nr, self.T).T                                            # (T, Nr) component means
        Vc = v.reshape(self.Nr, self.T).T
        mean = M @ self.V.T
        var = Vc @ (self.V.T ** 2) + sigma2_new
        return mean, var
