"""Host-side mirror of the reference's LGSSM interface, dispatching every T-step algorithm to the HIP
engine (libtgp_hip.so). Same names, argument meaning and error behaviour as

    /root/reference/src/models/lgssm.jl                 LGSSM, rand, marginals, logpdf, _filter, posterior
    /root/reference/src/models/gauss_markov_model.jl    Forward, Reverse, GaussMarkovModel
    /root/reference/src/models/linear_gaussian_conditionals.jl:225-257   ScalarOutputLGC
    /root/reference/src/models/missings.jl              missing handling, replace_observation_noise_cov
    /root/reference/src/util/gaussian.jl                Gaussian

Arrays follow NumPy conventions (A[t] is the d x d matrix, row-major); a leading dimension of 1 is a
FillArrays.Fill (one block shared by all steps -- what RegularSpacing inputs produce). Arrays may be
NumPy (host; copied to the device per call) or torch CUDA tensors (used in place, results stay on the
device). Missing observations are NaN entries of y (Julia: `missing`).
There is no CPU implementation here: without the built library and a GPU every call raises.
"""
import ctypes

import numpy as np

from . import _lib

__all__ = ["Forward", "Reverse", "Gaussian", "GaussMarkovModel", "SDETransitions", "ScalarOutputLGC", "SmallOutputLGC", "LGSSM", "logpdf", "_filter",
           "posterior", "marginals", "rand", "replace_observation_noise_cov", "posterior_marginals", "ε_randn", "logpdf_and_grad"]


class _Ordering:
    def __init__(self, name, code):
        self.name, self.code = name, code

    def __repr__(self):
        return self.name + "()"


Forward = _Ordering("Forward", 0)
Reverse = _Ordering("Reverse", 1)


def reverse(ordering):
    return Reverse if ordering is Forward else Forward


class Gaussian:
    """util/gaussian.jl:16-31."""

    def __init__(self, m, P):
        self.m, self.P = m, P

    def __repr__(self):
        return f"Gaussian(m={self.m}, P={self.P})"


def _is_torch(x):
    return hasattr(x, "data_ptr")


def _lead(x):
    return int(x.shape[0])


class GaussMarkovModel:
    """x[t] = A[t] x[t-1] + a[t] + eps[t], eps[t] ~ N(0, Q[t])  (gauss_markov_model.jl:20-32)."""

    def __init__(self, ordering, As, as_, Qs, x0):
        self.ordering, self.As, self.as_, self.Qs, self.x0 = ordering, As, as_, Qs, x0

    def __len__(self):
        return max(_lead(self.As), _lead(self.as_), _lead(self.Qs))

    @property
    def dim(self):
        return int(self.As.shape[-1])


class SDETransitions:
    """Transitions of an LTI SDE sampled at irregular times: A_k = exp(F dt_k), Q_k = P_inf - A_k P_inf A_k'
    (broadcast_components, lti_sde.jl:135-146, with the reference's dt_1 := 1), NOT materialised on the host:
    the device builds them from the time stamps (tgp_model_set_sde). x0.P doubles as P_inf."""

    def __init__(self, ordering, F, times, x0, A1=None, Q1=None):
        self.ordering, self.F, self.times, self.x0 = ordering, np.asarray(F, dtype=np.float64), np.asarray(times, dtype=np.float64), x0
        self.A1, self.Q1 = A1, Q1        # first transition (the reference takes dt_1 = 1 in each sub-kernel's own stretched time)

    def __len__(self):
        return len(self.times)

    @property
    def dim(self):
        return int(self.F.shape[0])


class ScalarOutputLGC:
    """StructArray of scalar-output emissions y | x ~ N(H'x + h, R) (lgc.jl:225-243): H (T|1, d), h (T|1,), R (T|1,)."""

    def __init__(self, H, h, R):
        self.H, self.h, self.R = H, h, R


class SmallOutputLGC:
    """StructArray of vector-output emissions y | x ~ N(H x + h, R) (lgc.jl:113-141): H (T|1, p, d), h (T|1, p),
    R (T|1, p) = the DIAGONAL of the noise covariance, or (T|1, p, p) dense. The engine absorbs the p
    observations of a time step as p scalar updates (exactly the joint update for diagonal noise); a dense R is
    whitened on the host first (H <- L^-1 H, h <- L^-1 h, y <- L^-1 y, R <- I), which supports logpdf / _filter /
    posterior; `marginals` (the diagonal, as for every vector-output model) and `rand` go through a diagonal-noise twin of the
    model, the correlated emission draw of `rand` being added on the host (`_diagonal_twin`)."""

    def __init__(self, H, h, R):
        self.H, self.h, self.R = H, h, R

    @property
    def p(self):
        return int(self.H.shape[-2])

    @property
    def dense(self):
        return self.R.ndim == 3


class LargeOutputLGC(SmallOutputLGC):
    """lgc.jl:146-217: the same conditional y | x ~ N(A x + a, Q) as SmallOutputLGC, which the reference evaluates through
    Cholesky factors of Q and P because dim_out > dim_in. On the device both are the same p scalar updates per time step
    (cost O(p d^2), no p x p factorisation at all), so this is SmallOutputLGC under the reference's name. The reference's
    Large form adds a 1e-10 jitter to P inside the update: results agree with it to the tolerance of its own
    consistency test (test/models/linear_gaussian_conditionals.jl:65-75)."""

    def __init__(self, A, a, Q):
        super().__init__(A, a, Q)


class BottleneckLGC(SmallOutputLGC):
    """lgc.jl:262-336: y | x ~ N(f.A (H x + h) + f.a, f.Q) with `fan_out` a LargeOutputLGC. The reference exploits the
    low-dimensional projection on the CPU; for the device the composition is folded into one vector-output emission
    (H <- f.A H, h <- f.A h + f.a, test/test_util.jl small_output_lgc_from_bottleneck), O(T p dz d) host work once.
    Agreement with the reference's own projected update: rtol 1e-6, its own test tolerance (:156-167)."""

    def __init__(self, H, h, fan_out):
        self.Hb, self.hb, self.fan_out = H, h, fan_out
        A, a = np.asarray(_to_numpy(fan_out.H), dtype=np.float64), np.asarray(_to_numpy(fan_out.h), dtype=np.float64)
        Hb, hb = np.asarray(_to_numpy(H), dtype=np.float64), np.asarray(_to_numpy(h), dtype=np.float64)
        super().__init__(np.matmul(A, Hb), np.einsum("...ij,...j->...i", A, hb) + a, fan_out.R)


class LGSSM:
    """lgssm.jl:9-12. `T` must be given when every array is a Fill."""

    def __init__(self, transitions, emissions, T=None, device=0):
        self.transitions, self.emissions = transitions, emissions
        n = max(len(transitions), _lead(emissions.H), _lead(emissions.h), _lead(emissions.R))
        if isinstance(transitions, SDETransitions):
            T = len(transitions)
        self.T = int(T) if T is not None else n
        if n > 1 and n != self.T:
            raise ValueError(f"inconsistent lengths: arrays have {n} steps, T={self.T}")
        self.device = device
        self._handle = None
        self._keep = None
        self._whiten = None      # (Linv (T|1,p,p), logdet_half (T|1,)) when a dense R was whitened
        self.handle_options = {}  # tgp_set_option values applied to the device handle before the model is bound

    @property
    def p(self):
        return self.emissions.p if isinstance(self.emissions, SmallOutputLGC) else 1

    def __len__(self):
        return self.T

    @property
    def ordering(self):
        return self.transitions.ordering

    @property
    def x0(self):
        return self.transitions.x0

    @property
    def dim(self):
        return self.transitions.dim

    # ---------------------------------------------------------------- device binding
    def _blocks(self, x, per, transpose=False):
        """-> (contiguous fp64 array in ABI layout, is_shared)."""
        shared = _lead(x) == 1
        if not shared and _lead(x) != self.T:
            raise ValueError(f"array has {_lead(x)} steps, expected 1 or {self.T}")
        if _is_torch(x):
            import torch
            t = x.to(torch.float64)
            if transpose:
                t = t.transpose(-1, -2)
            return t.contiguous(), shared
        a = np.asarray(x, dtype=np.float64)
        if transpose:
            a = np.swapaxes(a, -1, -2)
        return np.ascontiguousarray(a), shared

    def handle(self):
        """Create the device handle and upload / bind the model on first use."""
        if self._handle is not None:
            return self._handle
        tr, em = self.transitions, self.emissions
        d, p = self.dim, self.p
        if isinstance(tr, SDETransitions):
            return self._handle_sde()
        small = isinstance(em, SmallOutputLGC)
        eH, eh, eR = em.H, em.h, em.R
        if small and em.dense:
            # whiten: R = L L'  =>  H~ = L^-1 H, h~ = L^-1 h, R~ = I; logpdf gets - sum_t log det L_t (host arrays only)
            Rn = np.asarray(_to_numpy(eR), dtype=np.float64)
            Lc = np.linalg.cholesky(Rn)
            Linv = np.linalg.inv(Lc)
            eH = Linv @ np.asarray(_to_numpy(eH), dtype=np.float64)
            eh = (Linv @ np.asarray(_to_numpy(eh), dtype=np.float64)[..., None])[..., 0]
            n = max(eH.shape[0], eh.shape[0])
            eH = eH if eH.shape[0] == n else np.repeat(eH, n, axis=0)
            eh = eh if eh.shape[0] == n else np.repeat(eh, n, axis=0)
            eR = np.ones((1, p))
            self._whiten = (Linv, np.log(np.diagonal(Lc, axis1=-2, axis2=-1)).sum(axis=-1))
        A, sA = self._blocks(tr.As, d * d, transpose=True)   # column-major blocks == row-major of A'
        a, sa = self._blocks(tr.as_, d)
        Q, sQ = self._blocks(tr.Qs, d * d, transpose=True)
        H, sH = self._blocks(eH, p * d)       # (T|1, p, d) row-major: row j contiguous
        h, sh = self._blocks(eh, p)
        R, sR = self._blocks(eR, p)
        arrs = (A, a, Q, H, h, R)
        on_dev = [_lib.is_device(x) for x in arrs]
        if any(on_dev):
            _sync_torch(next(x for x in arrs if _lib.is_device(x)))
        if any(on_dev) and not all(on_dev):
            # SHARED blocks on the host beside per-step blocks on the device (a per-step noise that lives there, the blocks of an LTI model):
            # the few shared doubles follow to the device; a per-step host array beside device ones stays an error (T doubles would be uploaded)
            if any(not od and (x.numel() if _is_torch(x) else x.size) > 4096 for od, x in zip(on_dev, arrs)):
                raise ValueError("model arrays must be all NumPy or all CUDA tensors (shared blocks on the host may stand beside CUDA tensors)")
            import torch
            where = next(x for x in arrs if _lib.is_device(x)).device
            arrs = tuple(x if od else torch.as_tensor(x).to(where) for od, x in zip(on_dev, arrs))
            A, a, Q, H, h, R = arrs
            on_dev = [True] * len(arrs)
        flags = 0
        for bit, s in zip((_lib.SHARED_A, _lib.SHARED_a, _lib.SHARED_Q, _lib.SHARED_H, _lib.SHARED_h, _lib.SHARED_R),
                          (sA, sa, sQ, sH, sh, sR)):
            flags |= bit if s else 0
        if all(on_dev):
            flags |= _lib.DEVICE_PTRS
        if small:
            flags |= _lib.SMALL_OUTPUT
        hd = _lib.Handle(self.device)
        for opt, value in self.handle_options.items():
            hd.set_option(opt, value)
        x0m = np.ascontiguousarray(np.asarray(_to_numpy(self.x0.m), dtype=np.float64))
        x0P = np.ascontiguousarray(np.asarray(_to_numpy(self.x0.P), dtype=np.float64).T)
        hd.check(hd.lib.tgp_model_set(hd.h, self.T, d, p, self.ordering.code, flags, _lib.ptr(A), _lib.ptr(a), _lib.ptr(Q),
                                      _lib.ptr(H), _lib.ptr(h), _lib.ptr(R), _lib.ptr(x0m), _lib.ptr(x0P)))
        self._keep = arrs            # borrowed device pointers must outlive the handle
        self._handle = hd
        self._on_device = all(on_dev)
        return hd


def _sync_torch(t, model=None):
    """The library runs on its own HIP stream: make sure whatever torch has queued to produce a CUDA tensor
    we are about to read has finished (torch ops are asynchronous on torch's current stream).  (An event the handle's stream waits for on
    the device instead -- no host synchronisation -- was measured in round 5: the cross-stream dependency delays the kernel's start by
    ~10 us, three times what this synchronisation costs.)"""
    import torch
    raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    idx = t.device.index
    if raw_stream is None or idx is None:
        torch.cuda.current_stream(t.device).synchronize()
        return
    # (the raw hipStream_t, not a torch.cuda.Stream object: building one costs 1.7 us of a 40 us call -- scripts/py_overhead.py)
    if _lib.load().tgp_stream_synchronize(raw_stream(idx)) != _lib.OK:
        raise _lib.TGPError(_lib.EHIP, "hipStreamSynchronize on torch's current stream failed")


def _handle_sde(self):
    """Bind an SDE-described model: only F, P_inf, the emission blocks and the T time stamps go to the device."""
    tr, em = self.transitions, self.emissions
    if not isinstance(em, ScalarOutputLGC):
        raise NotImplementedError("SDE-described transitions support scalar observations only")
    d = tr.dim
    H, sH = self._blocks(em.H, d)
    h, sh = self._blocks(em.h, 1)
    R, sR = self._blocks(em.R, 1)
    flags = _lib.SHARED_a | (_lib.SHARED_H if sH else 0) | (_lib.SHARED_h if sh else 0) | (_lib.SHARED_R if sR else 0)
    hd = _lib.Handle(self.device)
    F = np.ascontiguousarray(tr.F.T)                                  # column-major
    a = np.zeros(d)
    times = np.ascontiguousarray(tr.times)
    x0m = np.ascontiguousarray(np.asarray(_to_numpy(self.x0.m), dtype=np.float64))
    x0P = np.ascontiguousarray(np.asarray(_to_numpy(self.x0.P), dtype=np.float64).T)
    A1 = None if tr.A1 is None else np.ascontiguousarray(np.asarray(tr.A1, dtype=np.float64).T)
    Q1 = None if tr.Q1 is None else np.ascontiguousarray(np.asarray(tr.Q1, dtype=np.float64).T)
    hd.check(hd.lib.tgp_model_set_sde(hd.h, self.T, d, self.ordering.code, flags, _lib.ptr(F), _lib.ptr(a), _lib.ptr(H), _lib.ptr(h),
                                      _lib.ptr(R), _lib.ptr(times), _lib.ptr(A1), _lib.ptr(Q1), _lib.ptr(x0m), _lib.ptr(x0P)))
    self._handle, self._on_device, self._keep = hd, False, (H, h, R)
    return hd


LGSSM._handle_sde = _handle_sde


def _to_numpy(x):
    if _is_torch(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _check_inputs(model, y):
    """lgssm.jl:202-208."""
    if len(model) != len(y):
        raise ValueError(f"Dimension mismatch. length(prior) is {len(model)}, but length(y) is {len(y)}")


_LAZY_NAN_MIN = 1 << 16


def _obs(y, model=None, lazy_nan=False):
    """-> (y contiguous fp64, missing mask or None, in_device). NaN == missing (host arrays only;
    for CUDA tensors pass a (y, mask) tuple). Vector observations: y (T, p); a (T,) mask marks whole steps.
    lazy_nan: a large scalar-output host series is NOT scanned for NaNs here (the scan costs more than its PCIe transfer: ~2-10 ms per
    1e7 values); a NaN reaches the device, comes back as a NaN log-likelihood, and the caller then calls again without lazy_nan
    (`_lazy_obs_call`). With lazy_nan the result has a fourth element: whether the scan was skipped."""
    if lazy_nan:
        return _obs_impl(y, model, True)
    return _obs_impl(y, model, False)[:3]


def _obs_impl(y, model, lazy_nan):
    mask = None
    if isinstance(y, tuple):
        y, mask = y
    if model is not None and model._whiten is not None and _is_torch(y) and y.is_cuda:
        # dense observation noise, observations on the device: whiten there (y <- L^-1 y), whole-step masks only
        import torch
        yy = y.to(torch.float64).reshape(model.T, model.p)
        mm = None if mask is None else mask.to(torch.bool)
        if mm is not None and mm.ndim == 2:
            if not bool(torch.all(mm.all(dim=1) | ~mm.any(dim=1))):
                raise TypeError("per-element missing observations need Diagonal noise (MethodError at lgc.jl:146)")
            mm = mm.all(dim=1)
        if mm is not None:
            yy = torch.where(mm[:, None], torch.zeros_like(yy), yy)
        Linv = torch.as_tensor(model._whiten[0], device=yy.device)
        yy = torch.matmul(Linv, yy[..., None])[..., 0].contiguous()
        mk = None if mm is None else mm[:, None].expand(model.T, model.p).to(torch.uint8).contiguous()
        _sync_torch(yy, model)
        return (yy if model.p > 1 else yy.reshape(model.T)), (mk if (mk is None or model.p > 1) else mk.reshape(model.T)), True, False
    if model is not None and (model.p > 1 or model._whiten is not None) and not (_is_torch(y) and y.is_cuda):
        yy = np.array(_to_numpy(y), dtype=np.float64)
        if model.p == 1 and yy.shape == (model.T,):
            yy = yy[:, None]
        if yy.shape != (model.T, model.p):
            raise ValueError(f"y must have shape ({model.T}, {model.p})")
        mk = np.isnan(yy) if mask is None else np.asarray(_to_numpy(mask), dtype=bool)
        if mk.ndim == 1:
            mk = np.repeat(mk[:, None], model.p, axis=1)
        if model._whiten is not None:
            if mk.any() and not np.all(mk.all(axis=1) | ~mk.any(axis=1)):
                raise TypeError("per-element missing observations need Diagonal noise (MethodError at lgc.jl:146)")
            Linv = model._whiten[0]
            yy = (Linv @ np.where(mk, 0.0, yy)[..., None])[..., 0]
        yy = np.ascontiguousarray(np.where(mk, 0.0, yy))
        if model.p == 1:
            yy, mk = yy.reshape(model.T), mk.reshape(model.T)
        return yy, (np.ascontiguousarray(mk.astype(np.uint8)) if mk.any() else None), False, False
    if _is_torch(y) and y.is_cuda:
        import torch
        yy = y if (y.dtype is torch.float64 and y.is_contiguous()) else y.to(torch.float64).contiguous()
        if mask is None:
            mm = None
        elif mask.dtype == torch.bool:
            mm = mask.contiguous().view(torch.uint8)      # (a bool tensor IS one byte per step, 0 / 1: no conversion kernel per call)
        else:
            mm = mask.to(torch.uint8).contiguous()
        _sync_torch(yy, model)
        return yy, mm, True, False
    if isinstance(y, np.ma.MaskedArray):
        mask = np.ma.getmaskarray(y) if mask is None else mask
        y = y.filled(0.0)
    yy = np.ascontiguousarray(_to_numpy(y), dtype=np.float64)
    if mask is None and lazy_nan and yy.size >= _LAZY_NAN_MIN:
        return yy, None, False, True
    if mask is None and np.isnan(yy).any():
        mask = np.isnan(yy)
    mm = None if mask is None else np.ascontiguousarray(_to_numpy(mask).astype(np.uint8))
    return yy, mm, False, False


def _lazy_obs_call(model, y, call):
    """call(yy, mm, dev) -> (lml, result). Large host series go to the device unscanned; a NaN log-likelihood (or a not-positive-
    definite report a NaN can cause) brings the NaN == missing scan back and, if it finds any, the call is repeated with the mask."""
    yy, mm, dev, unchecked = _obs(y, model, lazy_nan=True)
    if not unchecked:
        return call(yy, mm, dev)[1]
    try:
        lml, res = call(yy, mm, dev)
        if not np.isnan(lml):
            return res
    except _lib.NotPositiveDefinite:
        if not np.isnan(yy).any():
            raise
    yy, mm, dev = _obs(y, model)
    return call(yy, mm, dev)[1]


def _out(model, shape, like_device):
    if like_device:
        import torch
        return torch.empty(shape, dtype=torch.float64, device=f"cuda:{model.device}")
    return np.empty(shape, dtype=np.float64)


def _pair_statistic(y, R, ys, Rs):
    """Two observations y ~ N(f, R), y* ~ N(f, R*) of the same latent value with independent noise:
        N(y; f, R) N(y*; f, R*) = N(ybar; f, Rbar) N(y - y*; 0, R + R*),   Rbar = R R* / (R + R*),  ybar = Rbar (y / R + y* / R*)
    -> (ybar, Rbar, sum_t log N(y_t - y*_t; 0, R_t + R*_t)).  NaN (missing, missings.jl:25-33) on one side leaves the other observation as
    it is; on both sides the joint step is missing.  R, R*: one variance each (-> one Rbar unless something is missing) or one per step.
    Host arrays (the device counterpart is tgp_pair_statistic)."""
    y, ys = np.asarray(y, dtype=np.float64), np.asarray(ys, dtype=np.float64)
    R, Rs = np.atleast_1d(np.asarray(R, dtype=np.float64)), np.atleast_1d(np.asarray(Rs, dtype=np.float64))
    if R.shape[0] == 1 and Rs.shape[0] == 1 and y.shape[0] > 1 and np.isfinite(y.sum()) and np.isfinite(ys.sum()):
        # one variance on each side, nothing missing (a NaN would have made its sum one): four passes over the series instead of a dozen
        tot = float(R[0] + Rs[0])
        ybar = y - ys
        const = -0.5 * (y.shape[0] * np.log(2 * np.pi * tot) + float(ybar @ ybar) / tot)
        ybar *= Rs[0] / tot                          # ybar = y* + (y - y*) R* / (R + R*)
        ybar += ys
        return ybar, R * Rs / tot, const
    my, ms = np.isnan(y), np.isnan(ys)
    tot = R + Rs
    diff = y - ys
    both = ~(my | ms)
    const = -0.5 * float(np.sum((np.log(2 * np.pi * tot) + diff * diff / tot)[both] if tot.shape[0] > 1
                                else np.log(2 * np.pi * tot[0]) + diff[both] ** 2 / tot[0]))
    with np.errstate(invalid="ignore", divide="ignore"):     # (a zero variance on a side that is missing: the entry is replaced below)
        Rbar = R * Rs / tot
        ybar = (Rs * y + R * ys) / tot               # = Rbar (y / R + y* / R*) without the divisions by a tiny jitter R*
    if my.any() or ms.any():
        n = y.shape[0]
        Rbar = np.where(ms, np.broadcast_to(R, (n,)), np.where(my, np.broadcast_to(Rs, (n,)), np.broadcast_to(Rbar, (n,))))
        ybar = np.where(ms, y, np.where(my, ys, ybar))
    return ybar, Rbar, const


def _logpdf_with_noise(prior, ybar, Rbar):
    """logpdf of `prior` with its noise variance replaced by the ONE number Rbar, on the prior's own handle (tgp_logpdf_noise: no second model
    bound -- 0.09 ms of a 0.24 ms posterior logpdf at T = 1e7).  None: not a model of the one-launch paths; the caller binds the joint model."""
    if np.size(Rbar) != 1 or prior.T < 2 or isinstance(prior.transitions, SDETransitions):
        return None
    hd = prior.handle()
    out = ctypes.c_double()
    try:
        hd.check(hd.lib.tgp_logpdf_noise(hd.h, _lib.ptr(ybar), _lib.IN_DEVICE if _lib.is_device(ybar) else 0, float(np.reshape(_to_numpy(Rbar), -1)[0]),
                                         ctypes.byref(out)))
    except _lib.Unsupported:
        return None
    return out.value


def _posterior_logpdf_pair(post, y_new):
    """logpdf(replace_observation_noise_cov(posterior(prior, y), R_new), y_new) of a posterior that has not been evaluated, WITHOUT evaluating
    it: log p(y_new | y) = log p(y, y_new) - log p(y), and the two observations per step are one (`_pair_statistic` / tgp_pair_statistic) --
    two logpdf calls of the PRIOR on whatever engine it has instead of the filter of a T x (2 d^2 + d) reverse-time model.  Forward priors
    with scalar observations; None: the caller evaluates the posterior (lgssm.jl:193-221, then :147-151)."""
    prior = post._prior
    if (post._model is not None or isinstance(prior, PosteriorLGSSM) or prior.ordering is not Forward or prior.p != 1
            or prior._whiten is not None or not isinstance(prior.emissions, ScalarOutputLGC)):
        return None
    T = prior.T
    em = prior.emissions
    R = em.R
    Rn = post._R_new if post._R_new is not None else R
    on_dev = lambda v: _lib.is_device(v[0] if isinstance(v, tuple) else v)
    if on_dev(post._y) != on_dev(y_new):
        return None
    if not on_dev(y_new):
        # host arrays: NaN == missing straight into the statistic (no mask built and undone, no second NaN scan)
        def series(v):
            v, mask = v if isinstance(v, tuple) else (v, None)
            a = np.asarray(_to_numpy(v), dtype=np.float64)
            if a.shape != (T,):
                return None
            if mask is not None:
                a = a.copy()
                a[np.asarray(_to_numpy(mask), dtype=bool).reshape(T)] = np.nan
            return a
        yh, ynh = series(post._y), series(y_new)
        Rh, Rnh = np.atleast_1d(_to_numpy(R)).astype(np.float64).reshape(-1), np.atleast_1d(_to_numpy(Rn)).astype(np.float64).reshape(-1)
        if yh is None or ynh is None or Rh.shape[0] not in (1, T) or Rnh.shape[0] not in (1, T):
            return None
        ybar, Rbar, pair = _pair_statistic(yh, Rh, ynh, Rnh)
        if Rbar.shape[0] == 1 and T > 1:     # (one variance on each side and nothing missing: the joint model is the prior with another noise variance)
            lp = _logpdf_with_noise(prior, np.ascontiguousarray(ybar), Rbar)
            if lp is not None:
                return lp + pair - logpdf(prior, yh)
        joint = LGSSM(prior.transitions, ScalarOutputLGC(em.H, em.h, Rbar), T=T, device=prior.device)
        return logpdf(joint, ybar) + pair - logpdf(prior, yh)
    yy, my, dev = _obs(post._y, prior)
    yn, mn, devn = _obs(y_new, prior)
    if not (dev and devn) or tuple(yy.shape) != (T,) or tuple(yn.shape) != (T,):
        return None
    if dev:
        import torch
        hd = prior.handle()
        as_dev = lambda v: (v.to(torch.float64).reshape(-1).contiguous() if _is_torch(v) and v.is_cuda
                            else np.ascontiguousarray(np.atleast_1d(_to_numpy(v)), dtype=np.float64).reshape(-1))
        Rd, Rnd = as_dev(R), as_dev(Rn)
        to_dev = lambda v: v if _is_torch(v) or v.shape[0] == 1 else torch.as_tensor(v, device=yy.device)
        Rd, Rnd = to_dev(Rd), to_dev(Rnd)
        if Rd.shape[0] not in (1, T) or Rnd.shape[0] not in (1, T):
            return None
        shared = Rd.shape[0] == 1 and Rnd.shape[0] == 1 and my is None and mn is None and T > 1
        ybar = torch.empty(T, dtype=torch.float64, device=yy.device)
        Rbar = None if shared else torch.empty(T, dtype=torch.float64, device=yy.device)
        mbar = torch.empty(T, dtype=torch.uint8, device=yy.device) if (my is not None and mn is not None) else None
        pair = ctypes.c_double()
        _sync_torch(yy)
        hd.check(hd.lib.tgp_pair_statistic(hd.h, T, _lib.ptr(yy), _lib.ptr(my), _lib.ptr(Rd), int(Rd.shape[0]), _lib.ptr(yn), _lib.ptr(mn),
                                           _lib.ptr(Rnd), int(Rnd.shape[0]), _lib.ptr(ybar), _lib.ptr(Rbar), _lib.ptr(mbar), ctypes.byref(pair)))
        if shared:
            r, rn = float(_to_numpy(Rd)[0]), float(_to_numpy(Rnd)[0])
            Rbar = np.array([r * rn / (r + rn)])
        if shared:
            lp = _logpdf_with_noise(prior, ybar, Rbar)
            if lp is not None:
                return lp + pair.value - logpdf(prior, post._y)
        if _is_torch(Rbar) and (isinstance(prior.transitions, SDETransitions) or any(not _is_torch(b) and np.size(b) > 4096 for b in (
                prior.transitions.As, prior.transitions.as_, prior.transitions.Qs, em.H, em.h))):
            Rbar = _to_numpy(Rbar)       # (per-step blocks of the prior on the host, or time stamps: its noise joins them there)
        joint = LGSSM(prior.transitions, ScalarOutputLGC(em.H, em.h, Rbar), T=T, device=prior.device)
        return logpdf(joint, ybar if mbar is None else (ybar, mbar)) + pair.value - logpdf(prior, post._y)
    return None


def logpdf(model, y):
    """lgssm.jl:147-151 (+ missings.jl:8-13)."""
    if isinstance(model, PosteriorLGSSM) and model._model is None:
        lp = _posterior_logpdf_pair(model, y)
        if lp is not None:
            return lp
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    hd = model.handle()
    if model._whiten is None and model.p == 1:
        def call(yy, mm, dev):
            out = ctypes.c_double()
            hd.check(hd.lib.tgp_logpdf(hd.h, _lib.ptr(yy), _lib.ptr(mm), _lib.IN_DEVICE if dev else 0, ctypes.byref(out)))
            return out.value, out.value
        return _lazy_obs_call(model, y, call)
    yy, mm, dev = _obs(y, model)
    out = ctypes.c_double()
    hd.check(hd.lib.tgp_logpdf(hd.h, _lib.ptr(yy), _lib.ptr(mm), _lib.IN_DEVICE if dev else 0, ctypes.byref(out)))
    if model._whiten is not None:        # log N(y; ., S) = log N(L^-1 y; ., L^-1 S L^-T) - log det L  (observed steps only)
        ld = model._whiten[1]
        mh = None if mm is None else (mm.cpu().numpy() if _is_torch(mm) else mm).astype(bool)
        obs = np.ones(model.T, dtype=bool) if mh is None else ~mh.reshape(model.T, -1).all(axis=1)
        return out.value - float((ld if ld.shape[0] > 1 else np.repeat(ld, model.T))[obs].sum())
    return out.value


def logpdf_and_grad(model, y, tangents):
    """logpdf and d logpdf / d theta_k by forward-mode tangent scans on the device (tgp_logpdf_grad).
    `tangents`: list (one entry per parameter) of dicts with the derivatives of the SHARED model blocks:
    A (d,d), a (d,), Q (d,d), H (d,), h (), R (), x0m (d,), x0P (d,d) -- missing keys mean zero.
    The reference obtains this gradient by AD of the sequential loop (bench/single_output_gps.jl:149-156)."""
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    hd = model.handle()
    yy, mm, dev = _obs(y, model)
    d, n = model.dim, len(tangents)
    get = lambda t, k, shape: np.asarray(t.get(k, np.zeros(shape)), dtype=np.float64).reshape(shape)
    dA = np.ascontiguousarray(np.stack([get(t, "A", (d, d)).T for t in tangents]))       # column-major blocks
    dQ = np.ascontiguousarray(np.stack([get(t, "Q", (d, d)).T for t in tangents]))
    da = np.ascontiguousarray(np.stack([get(t, "a", (d,)) for t in tangents]))
    dH = np.ascontiguousarray(np.stack([get(t, "H", (d,)) for t in tangents]))
    dh = np.ascontiguousarray(np.array([float(get(t, "h", ())) for t in tangents]))
    dR = np.ascontiguousarray(np.array([float(get(t, "R", ())) for t in tangents]))
    dm = np.ascontiguousarray(np.stack([get(t, "x0m", (d,)) for t in tangents]))
    dP = np.ascontiguousarray(np.stack([get(t, "x0P", (d, d)).T for t in tangents]))
    lml, grad = ctypes.c_double(), np.zeros(n)
    hd.check(hd.lib.tgp_logpdf_grad(hd.h, _lib.ptr(yy), _lib.ptr(mm), _lib.IN_DEVICE if dev else 0, n, _lib.ptr(dA), _lib.ptr(da),
                                    _lib.ptr(dQ), _lib.ptr(dH), _lib.ptr(dh), _lib.ptr(dR), _lib.ptr(dm), _lib.ptr(dP),
                                    ctypes.byref(lml), _lib.ptr(grad)))
    return lml.value, grad


def logpdf_adjoint(model, y):
    """logpdf and its gradient with respect to the SHARED model blocks by ONE adjoint pass on the device (tgp_logpdf_adjoint): the cost
    of a posterior-marginals call whatever the number of hyper-parameters. Returns (lml, dict A (d,d), a (d,), Q (d,d), H (d,), h (),
    R (), x0m (d,), x0P (d,d)); Q and x0P gradients are symmetrised (pair them with symmetric tangents). Forward LTI models with one
    noise variance, scalar observations, no missing data, d <= 8 -- raises Unsupported otherwise (use logpdf_and_grad).
    The reference obtains this gradient by reverse-mode AD of the sequential loop (bench/single_output_gps.jl:149-156)."""
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    hd = model.handle()
    yy, mm, dev = _obs(y, model)
    if mm is not None:
        raise _lib.Unsupported(_lib.EUNSUPPORTED, "logpdf_adjoint: missing observations are not served by the adjoint pass")
    d = model.dim
    g = dict(A=np.zeros((d, d)), a=np.zeros(d), Q=np.zeros((d, d)), H=np.zeros(d), h=np.zeros(1), R=np.zeros(1), x0m=np.zeros(d), x0P=np.zeros((d, d)))
    lml = ctypes.c_double()
    hd.check(hd.lib.tgp_logpdf_adjoint(hd.h, _lib.ptr(yy), _lib.IN_DEVICE if dev else 0, ctypes.byref(lml),
                                       *[_lib.ptr(g[k]) for k in ("A", "a", "Q", "H", "h", "R", "x0m", "x0P")]))
    for k in ("A", "Q", "x0P"):
        g[k] = g[k].T.copy()          # column-major blocks -> [i][k]
    g["h"], g["R"] = float(g["h"][0]), float(g["R"][0])
    return lml.value, g


def logpdf_and_grad_sde(model, y, tangents, rel_step=1e-6):
    """The same for a model whose transitions are described by their SDE (SDETransitions: irregular spacing, d <= 4).
    `tangents`: per parameter, the derivatives of F (d,d), x0P == P_inf (d,d), x0m (d,), H (d,), h (), R () and of the explicit
    first transition A1, Q1 (d,d) -- missing keys mean zero. The per-step tangents dA_k, dQ_k are formed on the device
    (tgp_logpdf_grad_sde)."""
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    if not isinstance(model.transitions, SDETransitions):
        raise _lib.Unsupported(_lib.EUNSUPPORTED, "logpdf_and_grad_sde needs SDE-described transitions")
    hd = model.handle()
    yy, mm, dev = _obs(y, model)
    d, n = model.dim, len(tangents)
    get = lambda t, k, shape: np.asarray(t.get(k, np.zeros(shape)), dtype=np.float64).reshape(shape)
    colmaj = lambda key: np.ascontiguousarray(np.stack([get(t, key, (d, d)).T for t in tangents]))
    dF, dP, dA1, dQ1 = colmaj("F"), colmaj("x0P"), colmaj("A1"), colmaj("Q1")
    da = np.zeros((n, d))
    dH = np.ascontiguousarray(np.stack([get(t, "H", (d,)) for t in tangents]))
    dh = np.ascontiguousarray(np.array([float(get(t, "h", ())) for t in tangents]))
    dR = np.ascontiguousarray(np.array([float(get(t, "R", ())) for t in tangents]))
    dm = np.ascontiguousarray(np.stack([get(t, "x0m", (d,)) for t in tangents]))
    has1 = model.transitions.A1 is not None
    lml, grad = ctypes.c_double(), np.zeros(n)
    hd.check(hd.lib.tgp_logpdf_grad_sde(hd.h, _lib.ptr(yy), _lib.ptr(mm), _lib.IN_DEVICE if dev else 0, n, _lib.ptr(dF), _lib.ptr(dP),
                                        _lib.ptr(dA1) if has1 else None, _lib.ptr(dQ1) if has1 else None, _lib.ptr(da), _lib.ptr(dH),
                                        _lib.ptr(dh), _lib.ptr(dR), _lib.ptr(dm), _lib.ptr(dP), ctypes.c_double(rel_step),
                                        ctypes.byref(lml), _lib.ptr(grad)))
    return lml.value, grad


def _filter(model, y):
    """lgssm.jl:171-173: filtering distributions, returned as (means (T,d), covs (T,d,d))."""
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    hd = model.handle()
    yy, mm, dev = _obs(y, model)
    T, d = model.T, model.dim
    m, P = _out(model, (T, d), dev), _out(model, (T, d, d), dev)
    flags = (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0
    hd.check(hd.lib.tgp_filter(hd.h, _lib.ptr(yy), _lib.ptr(mm), flags, _lib.ptr(m), _lib.ptr(P), None))
    return m, P   # P blocks are symmetric, so the column-major blocks read correctly as row-major


def _materialise_posterior(model, y):
    """lgssm.jl:193-200 evaluated: the posterior LGSSM (opposite ordering, transitions (G, g, L), x0 = final filtering state)."""
    hd = model.handle()
    yy, mm, dev = _obs(y, model)
    T, d = model.T, model.dim
    G, g, L = _out(model, (T, d, d), dev), _out(model, (T, d), dev), _out(model, (T, d, d), dev)
    xfm, xfP = np.empty(d), np.empty((d, d))
    flags = (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0
    hd.check(hd.lib.tgp_posterior(hd.h, _lib.ptr(yy), _lib.ptr(mm), flags, _lib.ptr(G), _lib.ptr(g), _lib.ptr(L),
                                  _lib.ptr(xfm), _lib.ptr(xfP)))
    Gm = G.transpose(-1, -2) if dev else np.swapaxes(G, -1, -2)     # column-major blocks -> G[t][i, j]
    trans = GaussMarkovModel(reverse(model.ordering), Gm, g, L, Gaussian(xfm, xfP.T.copy()))
    return LGSSM(trans, model.emissions, T=T, device=model.device)


class PosteriorLGSSM:
    """What `posterior(prior, y)` returns: the posterior LGSSM of lgssm.jl:193-200, NOT YET EVALUATED. The reference's callers
    (posterior_lti_sde.jl:27-36, :50-58, :62-78) chain

        replace_observation_noise_cov(posterior(model, ys), S_new)  ->  marginals(.) | rand(rng, .) | logpdf(., ys_pr)

    `replace_observation_noise_cov` on this object only records the new noise, and `marginals` of the result runs the fused
    filter + RTS smoother (tgp_posterior_marginals: nothing of size T x (2 d^2 + d) is ever written). Anything else that
    looks inside -- transitions, x0, rand, logpdf, _filter, a further posterior -- evaluates the reverse-time model once
    (tgp_posterior) and then behaves as the plain LGSSM it stands for."""

    def __init__(self, prior, y, R_new=None):
        self._prior, self._y, self._R_new, self._model = prior, y, R_new, None
        self.T, self.device = prior.T, prior.device

    # -- lazy surface ------------------------------------------------------------------------------------
    @property
    def ordering(self):
        return reverse(self._prior.ordering)

    @property
    def p(self):
        return self._prior.p

    @property
    def dim(self):
        return self._prior.dim

    def __len__(self):
        return self.T

    def _emissions(self):
        em = self._prior.emissions
        if self._R_new is None:
            return em
        if isinstance(em, SmallOutputLGC):
            R = self._R_new if _is_torch(self._R_new) else np.asarray(self._R_new, dtype=np.float64)
            return SmallOutputLGC(em.H, em.h, R[None] if R.ndim == 1 else R)
        R = self._R_new if _is_torch(self._R_new) else np.atleast_1d(np.asarray(self._R_new, dtype=np.float64))
        return ScalarOutputLGC(em.H, em.h, R)

    @property
    def emissions(self):
        return self._emissions()

    # -- evaluation --------------------------------------------------------------------------------------
    def materialise(self):
        if self._model is None:
            post = _materialise_posterior(self._prior, self._y)
            self._model = LGSSM(post.transitions, self._emissions(), T=self.T, device=self.device)
        return self._model

    # what evaluates the reverse-time model when looked at (T x (2 d^2 + d) doubles through tgp_posterior): the fields and helpers of the
    # plain LGSSM it stands for -- and nothing else, so that a hasattr / getattr probe for an unrelated name stays an AttributeError
    # instead of a multi-GB evaluation
    _EVALUATED = frozenset({"transitions", "x0", "handle", "_whiten", "handle_options", "_on_device", "_handle", "_bound", "small_out"})

    def __getattr__(self, name):
        if name in PosteriorLGSSM._EVALUATED:
            return getattr(self.materialise(), name)
        raise AttributeError(f"{type(self).__name__!s} has no attribute {name!r}")

    def fused_marginals(self):
        """marginals(self) without evaluating the posterior model; None when only the evaluated route exists."""
        if self._model is not None:
            return None
        em = self._emissions()
        if isinstance(em, SmallOutputLGC) and em.dense:
            return None                 # dense new noise: the evaluated route (marginals of the materialised model)
        if self._prior.ordering is not Forward:
            return None                 # tgp_posterior_marginals smooths Forward priors only (lgssm.jl:223-228 is the evaluated route)
        pem = self._prior.emissions
        if isinstance(pem, SmallOutputLGC) and pem.dense:
            return None                 # a prior with dense observation noise is whitened on the host: evaluated route
        return posterior_marginals(self._prior, self._y, em.R)      # anything else that is unsupported is an ERROR, not a silent fallback


def posterior(model, y):
    """lgssm.jl:193-200: the posterior LGSSM (opposite ordering, transitions (G, g, L), x0 = final filtering state),
    returned unevaluated (PosteriorLGSSM)."""
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    return PosteriorLGSSM(model, y)


def replace_observation_noise_cov(model, R_new):
    """missings.jl:35-41. Vector observations: R_new (T|1, p) diagonal or (T|1, p, p) dense."""
    if isinstance(model, PosteriorLGSSM) and model._model is None:
        return PosteriorLGSSM(model._prior, model._y, R_new)
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    em = model.emissions
    if isinstance(em, SmallOutputLGC):
        R = R_new if _is_torch(R_new) else np.asarray(R_new, dtype=np.float64)
        if R.ndim == 1:
            R = R[None]
        return LGSSM(model.transitions, SmallOutputLGC(em.H, em.h, R), T=model.T, device=model.device)
    R = R_new if _is_torch(R_new) else np.atleast_1d(np.asarray(R_new, dtype=np.float64))
    return LGSSM(model.transitions, ScalarOutputLGC(em.H, em.h, R), T=model.T, device=model.device)


def _osh(model):
    return (model.T,) if model.p == 1 else (model.T, model.p)


def _need_diag(model, what):
    if model._whiten is not None or (isinstance(model.emissions, SmallOutputLGC) and model.emissions.dense):
        raise NotImplementedError(f"{what} with a dense observation-noise covariance is not implemented on the device "
                                  "(diagonal noise only)")


def _dense_noise(model):
    return isinstance(model, LGSSM) and isinstance(model.emissions, SmallOutputLGC) and model.emissions.dense


def _diagonal_twin(model):
    """For a model with a DENSE observation-noise covariance: the same transitions and emission maps with only diag(R) as noise, bound on
    the same device (cached on the model). Prior marginals and samples do not involve the Kalman update, so R enters them only
    additively: marginals_diag = diag(H P H') + diag(R) is what this twin's tgp_marginals returns, and rand = the twin's noise-free
    H x + h plus chol(R + 1e-9 I)' eps added on the host (lgc.jl:84-87)."""
    if getattr(model, "_twin", None) is None:
        em = model.emissions
        Rn = np.asarray(_to_numpy(em.R), dtype=np.float64)
        twin = LGSSM(model.transitions, SmallOutputLGC(em.H, em.h, np.ascontiguousarray(np.diagonal(Rn, axis1=-2, axis2=-1))), T=model.T,
                     device=model.device)
        twin.handle_options = dict(model.handle_options)
        model._twin = twin
    return model._twin


def marginals(model):
    """lgssm.jl:99-115: emission marginals of the model as given, returned as (mean (T,), var (T,)). On an unevaluated
    posterior this is the fused filter + RTS smoother (posterior_lti_sde.jl:27-36)."""
    if isinstance(model, PosteriorLGSSM):
        out = model.fused_marginals()
        if out is not None:
            return out
        model = model.materialise()
    if _dense_noise(model):
        return marginals(_diagonal_twin(model))      # (mean, DIAGONAL of H P H' + R): the marginals_diag contract for vector outputs
    _need_diag(model, "marginals")
    hd = model.handle()
    dev = model._on_device
    mean, var = _out(model, _osh(model), dev), _out(model, _osh(model), dev)
    hd.check(hd.lib.tgp_marginals(hd.h, _lib.OUT_DEVICE if dev else 0, _lib.ptr(mean), _lib.ptr(var)))
    return mean, var        # vector observations: var is the DIAGONAL of the p x p marginal covariance (marginals_diag)


def posterior_marginals(model, y, R_new, _with_lml=False, out=None):
    """marginals(replace_observation_noise_cov(posterior(model, y), R_new)) without materialising the
    posterior model -- the `marginals(posterior(fx, y)(x))` path (posterior_lti_sde.jl:27-36).
    out = (mean, var): result buffers of an earlier call to write into (a repeated call with the same device
    buffers is replayed from a recorded hipGraph, TGP_OPT_GRAPH)."""
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    _need_diag(model, "posterior_marginals")
    hd = model.handle()

    def call(yy, mm, dev):
        flags = (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0
        if dev and _is_torch(R_new):
            Rn = R_new.contiguous()
        elif dev:
            import torch
            Rn = torch.as_tensor(np.atleast_1d(np.asarray(R_new, dtype=np.float64)), device=yy.device)
        else:
            Rn = np.ascontiguousarray(np.atleast_1d(_to_numpy(R_new)), dtype=np.float64)
        if model.p > 1 and Rn.ndim == 1:
            Rn = Rn[None]
        if Rn.shape[0] == 1:
            flags |= _lib.SHARED_R
        elif Rn.shape[0] != model.T:
            raise ValueError("R_new must have length 1 or T")
        if model.p > 1 and tuple(Rn.shape[1:]) != (model.p,):
            raise ValueError(f"R_new must be (T|1, {model.p}) (diagonal of the new noise)")
        osh = _osh(model)
        mean, var = out if out is not None else (_out(model, osh, dev), _out(model, osh, dev))
        if out is not None and (_lib.is_device(mean) != bool(dev) or tuple(mean.shape) != osh or tuple(var.shape) != osh):
            raise ValueError("out: buffers of another call shape / memory space")
        # (the log marginal likelihood is always asked for: it is a by-product, and a NaN in it reports a NaN observation)
        lml = ctypes.c_double()
        hd.check(hd.lib.tgp_logpdf_and_posterior_marginals(hd.h, _lib.ptr(yy), _lib.ptr(mm), _lib.ptr(Rn), flags, ctypes.byref(lml),
                                                           _lib.ptr(mean), _lib.ptr(var)))
        return lml.value, ((lml.value, mean, var) if _with_lml else (mean, var))
    if model._whiten is None and model.p == 1:
        return _lazy_obs_call(model, y, call)
    return call(*_obs(y, model))[1]


def logpdf_and_posterior_marginals(model, y, R_new, out=None):
    """(logpdf(model, y), mean, var): logpdf and marginals(replace_observation_noise_cov(posterior(model, y), R_new)) of the
    same series from ONE forward filter + RTS smoother (tgp_logpdf_and_posterior_marginals) -- the log marginal likelihood
    is a by-product of the filter the posterior needs anyway. Diagonal noise (no whitening correction is applied here)."""
    if model._whiten is not None:
        raise NotImplementedError("logpdf_and_posterior_marginals with a dense observation-noise covariance")
    return posterior_marginals(model, y, R_new, _with_lml=True, out=out)


def posterior_marginals_at(model, y, H_new, h_new, R_new):
    """Marginals of N(H_new x_t + h_new, H_new P_t H_new' + R_new) under the SMOOTHED state x_t | y: what the reference gets
    by giving the posterior model other emissions (pseudo_point.jl:198-235) -- here without materialising that model
    (tgp_posterior_marginals_at). H_new (pn, d), h_new (pn,), R_new (T|1, pn) diagonal noise. Host arrays.
    Raises `_lib.Unsupported` where only the materialised route exists (d < 5, per-step transitions, Reverse models)."""
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    _check_inputs(model, y[0] if isinstance(y, tuple) else y)
    hd = model.handle()
    yy, mm, dev = _obs(y, model)
    Hn = np.ascontiguousarray(_to_numpy(H_new), dtype=np.float64)
    hn = np.ascontiguousarray(_to_numpy(h_new), dtype=np.float64)
    pn = int(Hn.shape[0])
    Rn = np.ascontiguousarray(np.atleast_2d(_to_numpy(R_new)), dtype=np.float64)
    if Hn.shape != (pn, model.dim) or hn.shape != (pn,) or Rn.shape[1] != pn or Rn.shape[0] not in (1, model.T):
        raise ValueError("H_new (pn, d), h_new (pn,), R_new (T|1, pn)")
    if dev:
        import torch
        Rn = torch.as_tensor(Rn, device=yy.device)
    flags = ((_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0) | (_lib.SHARED_R if Rn.shape[0] == 1 else 0)
    mean, var = _out(model, (model.T, pn), dev), _out(model, (model.T, pn), dev)
    hd.check(hd.lib.tgp_posterior_marginals_at(hd.h, _lib.ptr(yy), _lib.ptr(mm), pn, _lib.ptr(Hn), _lib.ptr(hn), _lib.ptr(Rn), flags,
                                               _lib.ptr(mean), _lib.ptr(var), None))
    return mean, var


def ε_randn(rng, model):
    """lgssm.jl:72-77: all the randomness one sample needs, drawn up front in the reference's order
    (T transition vectors, then T emission scalars; x0's draw comes after, lgssm.jl:67)."""
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    T, d = model.T, model.dim
    return rng.standard_normal((T, d)), rng.standard_normal(_osh(model))


def _posterior_rand_one_launch(post, eps_t, eps_e, eps_0):
    """rand of a posterior that has not been evaluated, on the prior's handle (tgp_posterior_rand: filter + reverse-time draw in one kernel,
    nothing of size T x (2 d^2 + d) written).  None: not a model of that path -- the caller evaluates the posterior."""
    prior = post._prior
    if (post._model is not None or isinstance(prior, PosteriorLGSSM) or prior.ordering is not Forward or prior.p != 1 or prior.dim > 6
            or prior._whiten is not None or isinstance(post._y, tuple)):
        return None
    y = post._y
    R_new = post._R_new if post._R_new is not None else prior.emissions.R
    dev = _lib.is_device(eps_t)
    if dev != _lib.is_device(y) or (not dev and np.isnan(np.asarray(_to_numpy(y), dtype=np.float64)).any()):
        return None
    hd = prior.handle()
    T, d = prior.T, prior.dim
    if dev:
        import torch
        # every buffer the kernel reads is float64, contiguous, of exactly the size it indexes, and on ONE device (round-5 advice: a float32 y or a
        # short eps_t would be read past its end)
        if not (_is_torch(eps_e) and _lib.is_device(eps_e)) or eps_t.numel() != T * d or y.numel() != T or eps_e.numel() != T:
            return None
        et = eps_t.to(torch.float64).reshape(T, d).contiguous()
        ee = eps_e.to(torch.float64).reshape(T).contiguous()
        yy = y.to(torch.float64).reshape(T).contiguous()
        if _is_torch(R_new):
            Rn = R_new.to(device=yy.device, dtype=torch.float64).reshape(-1).contiguous()
        else:
            Rn = torch.as_tensor(np.atleast_1d(np.asarray(_to_numpy(R_new), dtype=np.float64)).reshape(-1), device=yy.device)
        if et.device != yy.device or ee.device != yy.device:
            return None
        _sync_torch(et)
    else:
        if np.size(_to_numpy(eps_t)) != T * d or np.size(_to_numpy(y)) != T:
            return None
        et = np.ascontiguousarray(_to_numpy(eps_t), dtype=np.float64).reshape(T, d)
        ee = np.ascontiguousarray(_to_numpy(eps_e), dtype=np.float64)
        yy = np.ascontiguousarray(_to_numpy(y), dtype=np.float64).reshape(T)
        Rn = np.ascontiguousarray(np.atleast_1d(_to_numpy(R_new)), dtype=np.float64)
    if Rn.ndim != 1 or Rn.shape[0] not in (1, prior.T) or tuple(ee.shape) != (prior.T,):
        return None
    e0 = np.ascontiguousarray(_to_numpy(eps_0), dtype=np.float64).reshape(-1)
    if e0.shape[0] != d:
        return None
    out = _out(prior, _osh(prior), dev)
    flags = ((_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0) | (_lib.SHARED_R if Rn.shape[0] == 1 else 0)
    try:
        hd.check(hd.lib.tgp_posterior_rand(hd.h, _lib.ptr(yy), _lib.ptr(Rn), _lib.ptr(et), _lib.ptr(ee), _lib.ptr(e0), flags, _lib.ptr(out)))
    except _lib.Unsupported:
        return None
    return out


def rand(rng_or_eps, model):
    """lgssm.jl:65-69. `rng_or_eps` is a numpy Generator, or the explicit (eps_t (T,d), eps_e (T,), eps_0 (d,))."""
    if isinstance(model, PosteriorLGSSM) and model._model is None:
        if isinstance(rng_or_eps, tuple):
            eps = rng_or_eps
        else:       # (the reference's order of draws: lgssm.jl:72-77, then x0's)
            et, ee = rng_or_eps.standard_normal((model.T, model.dim)), rng_or_eps.standard_normal((model.T,) if model.p == 1 else (model.T, model.p))
            eps = (et, ee, rng_or_eps.standard_normal(model.dim))
        y1 = _posterior_rand_one_launch(model, *eps)
        if y1 is not None:
            return y1
        rng_or_eps = eps
    if isinstance(model, PosteriorLGSSM):
        model = model.materialise()
    if isinstance(rng_or_eps, tuple):
        eps_t, eps_e, eps_0 = rng_or_eps
    else:
        eps_t, eps_e = ε_randn(rng_or_eps, model)
        eps_0 = rng_or_eps.standard_normal(model.dim)
    if _dense_noise(model):
        # y_t = (H x_t + h) + chol(Symmetric(R_t + 1e-9 I)).U' eps_t (lgc.jl:84-87): the latent path and H x + h on the device with a zero
        # emission draw, the correlated noise (T small p x p products) on the host
        ee = np.asarray(_to_numpy(eps_e), dtype=np.float64).reshape(model.T, model.p)
        zero = np.zeros_like(ee)
        if _lib.is_device(eps_t):
            import torch
            zero = torch.zeros((model.T, model.p), dtype=torch.float64, device=eps_t.device)
        y0 = rand((eps_t, zero, eps_0), _diagonal_twin(model))
        Rn = np.asarray(_to_numpy(model.emissions.R), dtype=np.float64)
        Lc = np.linalg.cholesky(Rn + 1e-9 * np.eye(model.p))                       # (T|1, p, p) lower: U' = L
        noise = np.einsum("tij,tj->ti", np.broadcast_to(Lc, (model.T, model.p, model.p)), ee)
        if _is_torch(y0):
            import torch
            return y0 + torch.as_tensor(noise, device=y0.device).reshape(y0.shape)
        return y0 + noise.reshape(y0.shape)
    _need_diag(model, "rand")
    hd = model.handle()
    dev = _lib.is_device(eps_t)
    if dev:
        et, ee = eps_t.contiguous(), eps_e.contiguous()
        _sync_torch(et)
    else:
        et = np.ascontiguousarray(_to_numpy(eps_t), dtype=np.float64)
        ee = np.ascontiguousarray(_to_numpy(eps_e), dtype=np.float64)
    e0 = np.ascontiguousarray(_to_numpy(eps_0), dtype=np.float64)
    y = _out(model, _osh(model), dev)
    flags = (_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0
    hd.check(hd.lib.tgp_rand(hd.h, _lib.ptr(et), _lib.ptr(ee), _lib.ptr(e0), flags, _lib.ptr(y)))
    return y
