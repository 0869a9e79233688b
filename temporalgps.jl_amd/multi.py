"""ctypes twin of the in-library multi-GPU handle (include/tgp_hip.h: tgp_create_multi ...; csrc/tgp_multi.hip).

ONE process, one device handle + HIP stream + RCCL communicator per GPU inside libtgp_hip.so; the series is split
into contiguous time segments and the library runs the whole exchange itself -- this is the path a
caller that is not a torch.distributed job takes (the Julia glue: `DeviceLGSSM(...; ndev = 8)`). The reference has
no counterpart (its scan is the sequential loop of src/util/scan.jl:15-28).

`parallel.ShardedLGSSM` is the same protocol with one PROCESS per GPU and the collectives issued through
torch.distributed (what `torchrun bench.py` measures)."""
import ctypes

import numpy as np

from . import _lib
from . import lgssm as L

_vp = ctypes.c_void_p


def segment_bounds(T, ndev, rank):
    """[t0, t1) of `rank` (tgp_multi_segment; identical to parallel.segment_bounds)."""
    lib = _lib.load()
    t0, t1 = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.tgp_multi_segment(int(T), int(ndev), int(rank), ctypes.byref(t0), ctypes.byref(t1))
    if rc != _lib.OK:
        raise ValueError(f"tgp_multi_segment({T}, {ndev}, {rank}) -> {rc}")
    return t0.value, t1.value


class MultiHandle:
    """Owns one tgp_multi."""

    def __init__(self, devices):
        self.lib = _lib.load()
        devs = (ctypes.c_int * len(devices))(*[int(x) for x in devices])
        m = _vp()
        rc = self.lib.tgp_create_multi(ctypes.byref(m), len(devices), devs)
        if rc != _lib.OK:
            raise _lib.TGPError(rc, "tgp_create_multi failed (is a GPU visible? are the device ordinals valid?)")
        self.m = m
        self.devices = list(devices)

    def close(self):
        if getattr(self, "m", None):
            self.lib.tgp_destroy_multi(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def transport(self):
        return self.lib.tgp_multi_transport(self.m).decode()

    def check(self, rc):
        if rc == _lib.OK:
            return
        msg = self.lib.tgp_multi_last_error(self.m).decode()
        if rc == _lib.ENOTPD:
            raise _lib.NotPositiveDefinite(rc, msg)
        if rc == _lib.EUNSUPPORTED:
            raise _lib.Unsupported(rc, msg)
        if rc == _lib.EINVAL and "Dimension mismatch" in msg:
            raise ValueError(msg)
        raise _lib.TGPError(rc, msg)

    def set_option(self, opt, value):
        self.check(self.lib.tgp_multi_set_option(self.m, int(opt), int(value)))

    def rank_profile(self, rank):
        """per-kernel hipEvent profile of one rank's handle (TGP_OPT_PROFILE)"""
        h = self.lib.tgp_multi_handle(self.m, int(rank))
        out = {}
        buf = ctypes.create_string_buffer(128)
        for i in range(self.lib.tgp_profile_count(h)):
            ms, calls = ctypes.c_double(), ctypes.c_int64()
            self.lib.tgp_profile_get(h, i, buf, 128, ctypes.byref(ms), ctypes.byref(calls))
            out[buf.value.decode()] = dict(total_ms=ms.value, calls=calls.value)
        return out


def _parts(ptrs):
    """array of one pointer per rank (NULL entries allowed)"""
    return (_vp * len(ptrs))(*[_vp(p) if p else _vp(None) for p in ptrs])


class MultiLGSSM:
    """logpdf / posterior marginals of ONE series over `devices` (default: every visible GPU), host-array model.

    y, R_new: either whole-series host arrays (NumPy, sliced by pointer offset -- no copy) or a list with one CUDA
    tensor per rank holding that rank's segment on that rank's device; outputs come back the same way."""

    def __init__(self, model, devices=None):
        if not isinstance(model, L.LGSSM) or isinstance(model.transitions, L.SDETransitions):
            raise TypeError("MultiLGSSM takes an LGSSM with explicit blocks")
        if devices is None:
            import torch
            devices = list(range(torch.cuda.device_count()))
        self.model, self.devices, self.W = model, list(devices), len(devices)
        self.T, self.d, self.p = model.T, model.dim, model.p
        em, tr = model.emissions, model.transitions
        if isinstance(em, L.SmallOutputLGC) and em.dense:
            raise _lib.Unsupported(_lib.EUNSUPPORTED, "MultiLGSSM: dense observation noise is not sharded (whiten on the host first)")
        blk = lambda x, tp=False: model._blocks(np.asarray(L._to_numpy(x), dtype=np.float64), 0, transpose=tp)
        (A, sA), (a, sa), (Q, sQ) = blk(tr.As, True), blk(tr.as_), blk(tr.Qs, True)
        (H, sH), (h, sh), (R, sR) = blk(em.H), blk(em.h), blk(em.R)
        flags = 0
        for bit, s in zip((_lib.SHARED_A, _lib.SHARED_a, _lib.SHARED_Q, _lib.SHARED_H, _lib.SHARED_h, _lib.SHARED_R), (sA, sa, sQ, sH, sh, sR)):
            flags |= bit if s else 0
        if isinstance(em, L.SmallOutputLGC):
            flags |= _lib.SMALL_OUTPUT
        self.mh = MultiHandle(self.devices)
        for opt, value in model.handle_options.items():
            self.mh.set_option(opt, value)
        x0m = np.ascontiguousarray(np.asarray(L._to_numpy(model.x0.m), dtype=np.float64))
        x0P = np.ascontiguousarray(np.asarray(L._to_numpy(model.x0.P), dtype=np.float64).T)
        self.mh.check(self.mh.lib.tgp_multi_model_set(self.mh.m, self.T, self.d, self.p, model.ordering.code, flags, _lib.ptr(A), _lib.ptr(a),
                                                      _lib.ptr(Q), _lib.ptr(H), _lib.ptr(h), _lib.ptr(R), _lib.ptr(x0m), _lib.ptr(x0P)))
        self.bounds = [segment_bounds(self.T, self.W, r) for r in range(self.W)]

    @property
    def transport(self):
        return self.mh.transport

    # -- argument marshalling ------------------------------------------------------------------------------
    def _split(self, x, per_step_width, what):
        """-> (keepalive, [address per rank], on_device). Host: ONE contiguous array addressed at each segment's start."""
        if isinstance(x, (list, tuple)) and len(x) == self.W and all(L._is_torch(t) and t.is_cuda for t in x):
            import torch
            ts = []
            for r, t in enumerate(x):
                if t.device.index != self.devices[r]:
                    raise ValueError(f"{what}[{r}] lives on cuda:{t.device.index}, rank {r} runs on cuda:{self.devices[r]}")
                n = (self.bounds[r][1] - self.bounds[r][0]) * per_step_width
                if per_step_width and t.numel() != n:
                    raise ValueError(f"{what}[{r}] has {t.numel()} elements, the segment needs {n}")
                t = t.to(torch.float64).contiguous()
                L._sync_torch(t)
                ts.append(t)
            return ts, [t.data_ptr() for t in ts], True
        arr = np.ascontiguousarray(np.asarray(L._to_numpy(x), dtype=np.float64))
        if per_step_width and arr.size != self.T * per_step_width:
            raise ValueError(f"{what} has {arr.size} elements, expected {self.T * per_step_width}")
        base, w = arr.ctypes.data, per_step_width
        return arr, [base + 8 * self.bounds[r][0] * w for r in range(self.W)], False

    def _obs(self, y):
        mask = None
        if isinstance(y, tuple):
            y, mask = y
        keep, ptrs, dev = self._split(y, self.p, "y")
        mk = mptrs = None
        if not dev and mask is None and np.isnan(keep).any():
            mask = np.isnan(keep)
            keep = np.where(mask, 0.0, keep)
            ptrs = [keep.ctypes.data + 8 * self.bounds[r][0] * self.p for r in range(self.W)]
        if mask is not None:
            if dev:
                import torch
                if len(mask) != self.W:
                    raise ValueError(f"mask: one CUDA tensor per rank ({self.W}), got {len(mask)}")
                mk = [m.to(torch.uint8).contiguous() for m in mask]
                for r, m in enumerate(mk):
                    if m.numel() != (self.bounds[r][1] - self.bounds[r][0]) * self.p:
                        raise ValueError(f"mask of rank {r}: {m.numel()} values for a segment of {self.bounds[r][1] - self.bounds[r][0]} steps x {self.p}")
                mptrs = [m.data_ptr() for m in mk]
            else:
                mk = np.asarray(L._to_numpy(mask)).astype(np.uint8)
                if self.p > 1 and mk.shape == (self.T,):      # a whole-step mask of vector observations (as lgssm._obs accepts it)
                    mk = np.repeat(mk[:, None], self.p, axis=1)
                if mk.size != self.T * self.p:
                    raise ValueError(f"mask: {mk.size} values for a series of {self.T} steps x {self.p}")
                mk = np.ascontiguousarray(mk)
                mptrs = [mk.ctypes.data + self.bounds[r][0] * self.p for r in range(self.W)]
        return (keep, mk), ptrs, mptrs, dev

    def _rnew(self, R_new, dev):
        if isinstance(R_new, (list, tuple)) and dev:
            shared = all(t.numel() == self.p for t in R_new)
            keep, ptrs, _ = self._split(R_new, 0 if shared else self.p, "R_new")
            return keep, ptrs, shared
        arr = np.ascontiguousarray(np.asarray(L._to_numpy(R_new), dtype=np.float64)).reshape(-1)
        shared = arr.size == self.p and not (self.T == 1 and np.ndim(R_new) > 1 + (self.p > 1))      # (T == 1: a per-step array says so by its leading axis)
        if dev:      # a host scalar with device observations: one copy per rank
            import torch
            keep = [torch.as_tensor(arr, device=f"cuda:{self.devices[r]}") for r in range(self.W)]
            if not shared:
                raise ValueError("per-step R_new with device observations: pass one CUDA tensor per rank")
            return keep, [t.data_ptr() for t in keep], True
        if shared:
            return arr, [arr.ctypes.data] * self.W, True
        if arr.size != self.T * self.p:
            raise ValueError(f"R_new has {arr.size} elements, expected {self.p} or {self.T * self.p}")
        return arr, [arr.ctypes.data + 8 * self.bounds[r][0] * self.p for r in range(self.W)], False

    def _outputs(self, dev):
        shape = (self.T,) if self.p == 1 else (self.T, self.p)
        if dev:
            import torch
            mk = lambda: [torch.empty((b[1] - b[0],) + shape[1:], dtype=torch.float64, device=f"cuda:{self.devices[r]}") for r, b in enumerate(self.bounds)]
            mean, var = mk(), mk()
            return mean, var, [t.data_ptr() for t in mean], [t.data_ptr() for t in var]
        mean, var = np.empty(shape), np.empty(shape)
        off = lambda a: [a.ctypes.data + 8 * self.bounds[r][0] * self.p for r in range(self.W)]
        return mean, var, off(mean), off(var)

    # -- the calls -----------------------------------------------------------------------------------------
    def logpdf(self, y):
        """logpdf(model, y) of the whole series: lgssm.jl:147-151 (+ missings.jl:8-13)."""
        keep, ptrs, mptrs, dev = self._obs(y)
        out = ctypes.c_double()
        self.mh.check(self.mh.lib.tgp_multi_logpdf(self.mh.m, _parts(ptrs), None if mptrs is None else _parts(mptrs),
                                                   _lib.IN_DEVICE if dev else 0, ctypes.byref(out)))
        return out.value

    def _posterior(self, y, R_new, with_lml):
        keep, ptrs, mptrs, dev = self._obs(y)
        rkeep, rptrs, rshared = self._rnew(R_new, dev)
        mean, var, pm, pv = self._outputs(dev)
        flags = (_lib.IN_DEVICE | _lib.OUT_DEVICE if dev else 0) | (_lib.SHARED_R if rshared else 0)
        lml = ctypes.c_double()
        lib, mp = self.mh.lib, None if mptrs is None else _parts(mptrs)
        if with_lml:
            self.mh.check(lib.tgp_multi_logpdf_and_posterior_marginals(self.mh.m, _parts(ptrs), mp, _parts(rptrs), flags, ctypes.byref(lml), _parts(pm), _parts(pv)))
            return lml.value, mean, var
        self.mh.check(lib.tgp_multi_posterior_marginals(self.mh.m, _parts(ptrs), mp, _parts(rptrs), flags, _parts(pm), _parts(pv)))
        return mean, var

    def posterior_marginals(self, y, R_new):
        """marginals(posterior(model, y) with emission noise R_new): lgssm.jl:193-200 followed by :111-115."""
        return self._posterior(y, R_new, False)

    def logpdf_and_posterior_marginals(self, y, R_new):
        return self._posterior(y, R_new, True)
