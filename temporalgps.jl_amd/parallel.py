"""Time-sharding of one LGSSM across the GPUs of a node (not in the reference, whose
scan is a single sequential loop, /root/reference/src/util/scan.jl:15-28).

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI, "gloo" in the CPU tests). Rank r
owns the contiguous segment [r T/W, (r+1) T/W) of the series and an LGSSM over just that segment.

  forward:   every rank reduces its segment to ONE filter element           (tgp_segment_reduce, local)
             all_gather of the W elements (<= a few hundred bytes each)       -- the only forward exchange
             rank r folds elements 0..r-1 onto the prior x0 -> its carry-in   (tgp_elem_apply, host, W-1 tiny ops)
             local filter from the carry-in (reusing pass 1), all_reduce(sum) of one scalar for logpdf
  backward:  every rank reduces its segment to ONE smoother element          (tgp_smoother_forward, local)
             all_gather of the W elements + the last rank's final state       -- the only backward exchange
             rank r folds elements W-1..r+1 onto x_T|T -> smoothed state at its segment end, then smooths locally

No bulk data ever crosses xGMI: the exchange is latency-bound, outputs stay sharded.

Two transports for the same protocol:
  * device-resident (backend "nccl" == RCCL, the product path on GPUs): the elements never leave HBM. The
    tgp_shard_* entry points only ENQUEUE on one HIP stream, the all-gathers / the all-reduce are RCCL
    collectives ordered on that stream, k_fold applies the gathered elements on the device; the host
    synchronises ONCE per call (to read 4 doubles for logpdf, or when the smoother has finished).
  * host (backend "gloo": the CPU tests, and multi-rank tests on a single GPU): elements are copied to the
    host, exchanged with gloo and folded with tgp_elem_apply.
"""
import ctypes

import numpy as np

from . import _lib
from . import lgssm as L


def segment_bounds(T, world, rank):
    """Contiguous, balanced split of 0..T into `world` segments. Long series: interior boundaries on multiples of 512 steps (a shard
    of the stationary-gain engine that hands its end state on must be a whole number of that engine's tiles) -- the rule of
    tgp_multi_segment (csrc/tgp_multi.hip)."""
    T, world = int(T), int(world)
    base, rem = divmod(T, world)

    def lo(r):
        if r <= 0:
            return 0
        if r >= world:
            return T
        v = r * base + min(r, rem)
        return v // 512 * 512 if base >= 8 * 512 else v
    return lo(rank), lo(rank + 1)


class HIPEngine:
    """The product engine: thin calls into libtgp_hip.so for one segment model."""

    def __init__(self, model):
        self.model, self.hd = model, model.handle()
        self.d = model.dim

    def _obs(self, y):
        yy, mm, dev = L._obs(y)
        return yy, mm, (_lib.IN_DEVICE if dev else 0), dev

    def x0(self):
        return np.asarray(L._to_numpy(self.model.x0.m), dtype=np.float64), np.asarray(L._to_numpy(self.model.x0.P), dtype=np.float64)

    def elem_size(self, kind):
        return self.hd.lib.tgp_elem_size(kind, self.d)

    def segment_reduce(self, y):
        yy, mm, fl, _ = self._obs(y)
        out = np.empty(self.elem_size(0))
        self.hd.check(self.hd.lib.tgp_segment_reduce(self.hd.h, _lib.ptr(yy), _lib.ptr(mm), fl, _lib.ptr(out)))
        return out

    def elem_apply(self, kind, elem, m, P):
        mo, Po = np.empty(self.d), np.empty((self.d, self.d))
        Pc = np.ascontiguousarray(np.asarray(P, dtype=np.float64).T)
        e = np.ascontiguousarray(elem, dtype=np.float64)
        mc = np.ascontiguousarray(m, dtype=np.float64)
        rc = self.hd.lib.tgp_elem_apply(kind, self.d, _lib.ptr(e), _lib.ptr(mc), _lib.ptr(Pc), _lib.ptr(mo), _lib.ptr(Po))
        if rc != 0:
            raise _lib.TGPError(rc, "tgp_elem_apply")
        return mo, Po.T.copy()

    def set_x0(self, m, P):
        mc = np.ascontiguousarray(m, dtype=np.float64)
        Pc = np.ascontiguousarray(np.asarray(P, dtype=np.float64).T)
        self.hd.check(self.hd.lib.tgp_model_set_x0(self.hd.h, _lib.ptr(mc), _lib.ptr(Pc)))

    def logpdf(self, y, reuse):
        yy, mm, fl, _ = self._obs(y)
        out = ctypes.c_double()
        self.hd.check(self.hd.lib.tgp_logpdf(self.hd.h, _lib.ptr(yy), _lib.ptr(mm), fl | (_lib.REUSE_REDUCE if reuse else 0),
                                             ctypes.byref(out)))
        return out.value

    def smoother_forward(self, y, reuse):
        yy, mm, fl, _ = self._obs(y)
        rev = np.empty(self.elem_size(1))
        xfm, xfP = np.empty(self.d), np.empty((self.d, self.d))
        lml = ctypes.c_double()
        self.hd.check(self.hd.lib.tgp_smoother_forward(self.hd.h, _lib.ptr(yy), _lib.ptr(mm), fl | (_lib.REUSE_REDUCE if reuse else 0),
                                                       _lib.ptr(rev), _lib.ptr(xfm), _lib.ptr(xfP), ctypes.byref(lml)))
        return rev, xfm, xfP.T.copy(), lml.value

    # -- device-resident exchange (tgp_shard_*): everything below only enqueues on self.stream()
    def stream(self):
        """The HIP stream the handle and this rank's collectives run on (created on first use)."""
        import torch
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream(device=torch.device("cuda", self.model.device))
            self.hd.check(self.hd.lib.tgp_set_stream(self.hd.h, ctypes.c_void_p(self._stream.cuda_stream)))
        return self._stream

    def slot_size(self, phase):
        return self.hd.lib.tgp_shard_slot_size(phase, self.d)

    def device_obs(self, y):
        """y (and an optional mask) as contiguous CUDA tensors; no host synchronisation."""
        import torch
        mask = None
        if isinstance(y, tuple):
            y, mask = y
        dev = torch.device("cuda", self.model.device)
        if isinstance(y, np.ma.MaskedArray):
            mask = np.ma.getmaskarray(y) if mask is None else mask
            y = y.filled(0.0)
        if not L._is_torch(y):
            y = np.asarray(y, dtype=np.float64)
            if mask is None and np.isnan(y).any():
                mask = np.isnan(y)
        yy = torch.as_tensor(y, dtype=torch.float64, device=dev).contiguous()
        mm = None if mask is None else torch.as_tensor(np.asarray(mask) if not L._is_torch(mask) else mask, device=dev).to(torch.uint8).contiguous()
        if yy.shape[0] != self.model.T:
            raise ValueError(f"y has {yy.shape[0]} entries, the segment model has {self.model.T}")
        return yy, mm

    def shard_reduce(self, yy, mm, slot):
        self.hd.check(self.hd.lib.tgp_shard_reduce(self.hd.h, _lib.ptr(yy), _lib.ptr(mm), _lib.IN_DEVICE, _lib.ptr(slot)))

    def shard_fold(self, gathered, world, rank):
        self.hd.check(self.hd.lib.tgp_shard_fold(self.hd.h, _lib.ptr(gathered), int(world), int(rank)))

    def shard_logpdf(self, stats):
        self.hd.check(self.hd.lib.tgp_shard_logpdf(self.hd.h, _lib.ptr(stats)))

    def shard_smoother_forward(self, slot):
        self.hd.check(self.hd.lib.tgp_shard_smoother_forward(self.hd.h, _lib.ptr(slot)))

    def shard_smoother_backward(self, gathered, world, rank, R_new, out_device):
        import torch
        dev = torch.device("cuda", self.model.device)
        T = self.model.T
        Rn = torch.as_tensor(np.atleast_1d(np.asarray(R_new, dtype=np.float64)) if not L._is_torch(R_new) else R_new,
                             dtype=torch.float64, device=dev).reshape(-1).contiguous()
        mean, var = L._out(self.model, (T,), out_device), L._out(self.model, (T,), out_device)
        flags = _lib.IN_DEVICE | (_lib.OUT_DEVICE if out_device else 0) | (_lib.SHARED_R if Rn.shape[0] == 1 else 0)
        lml = ctypes.c_double()
        self.hd.check(self.hd.lib.tgp_shard_smoother_backward(self.hd.h, _lib.ptr(gathered), int(world), int(rank), _lib.ptr(Rn), flags,
                                                              _lib.ptr(mean), _lib.ptr(var), ctypes.byref(lml)))
        self.last_segment_lml = lml.value      # this segment's share of the log marginal likelihood (by-product of pass 2)
        return mean, var

    def smoother_backward(self, xs, R_new, like):
        dev = _lib.is_device(like)
        T = self.model.T
        mean, var = L._out(self.model, (T,), dev), L._out(self.model, (T,), dev)
        if dev and not L._is_torch(R_new):
            import torch
            R_new = torch.as_tensor(np.atleast_1d(np.asarray(R_new, dtype=np.float64)), device=like.device)
        Rn = R_new.contiguous() if L._is_torch(R_new) else np.ascontiguousarray(np.atleast_1d(R_new), dtype=np.float64)
        flags = ((_lib.IN_DEVICE | _lib.OUT_DEVICE) if dev else 0) | (_lib.SHARED_R if Rn.shape[0] == 1 else 0)
        if xs is None:
            pm = pP = None
        else:
            pm = np.ascontiguousarray(xs[0], dtype=np.float64)
            pP = np.ascontiguousarray(np.asarray(xs[1], dtype=np.float64).T)
        self.hd.check(self.hd.lib.tgp_smoother_backward(self.hd.h, _lib.ptr(pm), _lib.ptr(pP), _lib.ptr(Rn), flags,
                                                        _lib.ptr(mean), _lib.ptr(var)))
        return mean, var


class DistComm:
    """torch.distributed collectives on device tensors (RCCL): asynchronous, ordered on the current stream."""

    def __init__(self, group=None):
        self.group = group

    def all_gather(self, gathered, slot):
        import torch.distributed as dist
        dist.all_gather_into_tensor(gathered, slot, group=self.group)

    def all_reduce_sum(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


class ShardedLGSSM:
    """logpdf / posterior marginals of a series whose time axis is split across `world` ranks.
    `model` is this rank's segment LGSSM (x0 = the GLOBAL prior on every rank). With world == 1 this is
    exactly the single-GPU path. `engine` is injectable for the CPU (gloo) tests."""

    def __init__(self, model, world=1, rank=0, engine=None, group=None, comm=None):
        self.model, self.world, self.rank, self.group = model, int(world), int(rank), group
        self.engine = engine if engine is not None else (HIPEngine(model) if self.world > 1 else None)
        self.comm = comm          # device-resident transport; None => torch.distributed (chosen by backend)

    @property
    def transport(self):
        """'device' (elements stay in HBM, RCCL collectives on the handle's stream), 'host', or 'none' (single GPU)."""
        if self.engine is None:
            return "none"
        return "device" if self._device_resident() else "host"

    # -- device-resident transport -----------------------------------------------------------------------
    def _device_resident(self):
        if not isinstance(self.engine, HIPEngine) or getattr(self, "_dx_disabled", False):
            return False
        if self.comm is not None:
            return True
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl":
            self.comm = DistComm(self.group)
            return True
        return False

    def _buffers(self):
        if not hasattr(self, "_slot"):
            import torch
            e, dev = self.engine, torch.device("cuda", self.model.device)
            n0, n1 = e.slot_size(0), e.slot_size(1)
            mk = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
            self._slot = (mk(n0), mk(n1))
            self._gath = (mk(self.world * n0), mk(self.world * n1))
            self._stats = mk(4)
        return self._slot, self._gath, self._stats

    def _forward_device(self, y):
        """enqueue pass 1, the all-gather of the filter elements and the carry-in fold"""
        e = self.engine
        slot, gath, _ = self._buffers()
        yy, mm = e.device_obs(y)
        e.shard_reduce(yy, mm, slot[0])
        self.comm.all_gather(gath[0], slot[0])
        e.shard_fold(gath[0], self.world, self.rank)
        return yy

    def _one_launch_device(self, y, R_new, post):
        """The call on the one-launch path's time segments (tgp_segment_*, round 4): both mean recursions of the stationary region forget a
        state within `halo` steps, so a rank needs nothing of its neighbours but their `halo` observations next to the boundary -- ONE
        all-gather of 2 halo observations per rank, the segment's single kernel, and the sum of the ranks' shares of the log marginal
        likelihood.  Whether it applies is a function of the model blocks and the segment bounds alone: every rank evaluates it for every
        segment and reaches the same verdict without communication.  Data that only some ranks see (a NaN == missing value) surfaces as
        a NaN total, which every rank sees: all of them then take the general protocol together.  Returns None when it does not apply,
        else (share, mean, var) with shares that add up to the log marginal likelihood.  Must be called inside the handle's stream context."""
        import torch
        e = self.engine
        lib, hd = e.hd.lib, e.hd
        if getattr(self, "_one_off", False) or self.model.p != 1 or isinstance(y, tuple):
            return None
        dev = torch.device("cuda", self.model.device)
        if not hasattr(self, "_geom"):
            mine = torch.tensor([float(self.model.T)], dtype=torch.float64, device=dev)
            allT = torch.zeros(self.world, dtype=torch.float64, device=dev)
            self.comm.all_gather(allT, mine)
            Ts = [int(v) for v in allT.cpu().tolist()]          # (one synchronisation, once per bound model)
            self._geom = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int64)
            self._halo = None
        bounds = self._geom
        T, lo, hi = int(bounds[-1]), int(bounds[self.rank]), int(bounds[self.rank + 1])
        ok, halo = ctypes.c_int32(0), ctypes.c_int32(0)
        hd.check(lib.tgp_segment_plan(hd.h, T, self.world, bounds.ctypes.data, ctypes.byref(ok), ctypes.byref(halo)))
        if not ok.value:
            self._one_off = True          # (the same verdict on every rank)
            return None
        H = int(halo.value)
        yy, _ = e.device_obs(y)          # (a NaN == missing value stays in: it surfaces as a NaN total below)
        if self._halo != H:
            self._edge = torch.zeros(2 * H, dtype=torch.float64, device=dev)
            self._edges = torch.zeros(self.world * 2 * H, dtype=torch.float64, device=dev)
            self._tot = torch.zeros(1, dtype=torch.float64, device=dev)
            self._halo = H
        self._edge[:H].copy_(yy[:H])
        self._edge[H:].copy_(yy[-H:])
        self.comm.all_gather(self._edges, self._edge)
        left = self._edges[(self.rank - 1) * 2 * H + H:(self.rank - 1) * 2 * H + 2 * H] if self.rank > 0 else None
        right = self._edges[(self.rank + 1) * 2 * H:(self.rank + 1) * 2 * H + H] if self.rank < self.world - 1 else None
        out_device = L._is_torch(y)
        mean = var = Rn = None
        flags = _lib.IN_DEVICE
        if post:
            Rn = torch.as_tensor(np.atleast_1d(np.asarray(R_new, dtype=np.float64)) if not L._is_torch(R_new) else R_new,
                                 dtype=torch.float64, device=dev).reshape(-1).contiguous()
            if Rn.numel() not in (1, self.model.T):      # (the C side reads T values of a per-step array: never hand it fewer)
                raise ValueError(f"R_new must hold one value or one per step of this rank's segment ({self.model.T}), got {Rn.numel()}")
            mean, var = L._out(self.model, (self.model.T,), out_device), L._out(self.model, (self.model.T,), out_device)
            flags |= (_lib.OUT_DEVICE if out_device else 0) | (_lib.SHARED_R if Rn.shape[0] == 1 else 0)
        share = ctypes.c_double()
        hd.check(lib.tgp_segment_logpdf_and_posterior_marginals(hd.h, T, lo, hi, _lib.ptr(yy), _lib.ptr(left), _lib.ptr(right), _lib.ptr(Rn), flags,
                                                                _lib.ptr(mean), _lib.ptr(var), ctypes.byref(share)))        # the rank's synchronisation
        self._tot[0] = share.value
        self.comm.all_reduce_sum(self._tot)
        total = float(self._tot.cpu()[0])
        if not np.isfinite(total):            # (every rank sees the same total)
            self._one_off = True
            return None
        return total / self.world, mean, var

    def _steady_device(self, y, R_new, post):
        res = self._one_launch_device(y, R_new, post)
        if res is not None:
            return res
        return self._steady_shards_device(y, R_new, post)

    def _steady_shards_device(self, y, R_new, post):
        """The same call on the stationary-gain engine's shards (LTI models; tgp_shard_steady_begin / _finish: two halves around ONE
        all-gather). Returns None when it does not apply -- agreed by every rank through the gathered elements, so that all of them
        then take the general protocol together. Must be called inside the handle's stream context."""
        import torch
        e = self.engine
        lib, hd = e.hd.lib, e.hd
        n = lib.tgp_shard_steady_slot_size(e.d)
        if n == 0 or self.model.p != 1 or getattr(self, "_steady_off", False) or isinstance(y, tuple):
            return None
        if not hasattr(self, "_sslot"):
            dev = torch.device("cuda", self.model.device)
            self._sslot = torch.zeros(n, dtype=torch.float64, device=dev)
            self._sgath = torch.zeros(self.world * n, dtype=torch.float64, device=dev)
        yy, mm = e.device_obs(y)
        began = False
        if mm is None:
            rc = lib.tgp_shard_steady_begin(hd.h, _lib.ptr(yy), _lib.IN_DEVICE, int(self.rank == 0), int(self.rank == self.world - 1), int(post),
                                            _lib.ptr(self._sslot))
            began = rc == _lib.OK
            if not began and rc != _lib.EUNSUPPORTED:
                hd.check(rc)
        # A segment with missing data (NaN == missing: a property of THIS rank's slice of a host series, which the other ranks cannot see)
        # or one the engine declines takes part in the gather all the same, with a zero element: "does not apply" travels with the element,
        # every rank learns it from the gathered words and all of them take the general protocol together.
        if not began:
            self._sslot.zero_()
        self.comm.all_gather(self._sgath, self._sslot)
        if not began:
            self._steady_off = True
            return None
        out_device = L._is_torch(y)
        mean = var = Rn = None
        flags = _lib.IN_DEVICE
        if post:
            dev = torch.device("cuda", self.model.device)
            Rn = torch.as_tensor(np.atleast_1d(np.asarray(R_new, dtype=np.float64)) if not L._is_torch(R_new) else R_new,
                                 dtype=torch.float64, device=dev).reshape(-1).contiguous()
            mean, var = L._out(self.model, (self.model.T,), out_device), L._out(self.model, (self.model.T,), out_device)
            flags |= (_lib.OUT_DEVICE if out_device else 0) | (_lib.SHARED_R if Rn.shape[0] == 1 else 0)
        lml, served = ctypes.c_double(), ctypes.c_int(0)
        hd.check(lib.tgp_shard_steady_finish(hd.h, _lib.ptr(self._sgath), self.world, self.rank, _lib.ptr(Rn), flags, _lib.ptr(mean), _lib.ptr(var),
                                             ctypes.byref(lml), ctypes.byref(served)))        # the rank's one synchronisation
        if not served.value:
            self._steady_off = True
            return None
        return lml.value, mean, var

    def _logpdf_device(self, y):
        import torch
        e = self.engine
        st = e.stream()
        st.wait_stream(torch.cuda.current_stream(st.device))
        with torch.cuda.stream(st):
            res = self._steady_device(y, None, False)
            if res is not None:
                stats = self._buffers()[2]
                stats.zero_()
                stats[0] = res[0]
                self.comm.all_reduce_sum(stats)
                return float(stats.cpu()[0])
            self._forward_device(y)
            stats = self._buffers()[2]
            e.shard_logpdf(stats)
            self.comm.all_reduce_sum(stats)
            s = stats.cpu()                                    # the one synchronisation of the call
        if s[2] != 0 or s[3] != 0:
            raise _lib.NotPositiveDefinite(_lib.ENOTPD, "innovation variance / predicted covariance not positive definite (some shard)")
        return float(s[0])

    def _posterior_marginals_device(self, y, R_new):
        import torch
        e = self.engine
        st = e.stream()
        st.wait_stream(torch.cuda.current_stream(st.device))
        with torch.cuda.stream(st):
            res = self._steady_device(y, R_new, True)
            if res is not None:
                e.last_segment_lml = res[0]
                return res[1], res[2]
            yy = self._forward_device(y)
            slot, gath, _ = self._buffers()
            e.shard_smoother_forward(slot[1])
            self.comm.all_gather(gath[1], slot[1])
            out_device = L._is_torch(y[0] if isinstance(y, tuple) else y)
            mean, var = e.shard_smoother_backward(gath[1], self.world, self.rank, R_new, out_device)   # synchronises
        return mean, var

    # -- collectives on tiny host vectors (the payload is a few hundred bytes; latency-bound)
    def _all_gather(self, vec):
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.as_tensor(np.ascontiguousarray(vec), dtype=torch.float64).to(dev)
        try:                                      # one flat buffer => ONE device-to-host copy for all W elements
            flat = torch.empty(self.world * t.numel(), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(flat, t, group=self.group)
            return list(flat.cpu().numpy().reshape(self.world, -1))
        except (RuntimeError, AttributeError, NotImplementedError):
            out = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(out, t, group=self.group)
            return [o.cpu().numpy() for o in out]

    def _all_reduce_sum(self, x):
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def _forward_exchange(self, y):
        """Pass 1 + exchange: sets this segment's carry-in state. Always recomputed (no caching across API
        calls: each logpdf / posterior_marginals call does its full work, like the reference's); the True it
        returns lets the SAME call reuse the pass-1 elements it has just produced."""
        e = self.engine
        elems = self._all_gather(e.segment_reduce(y))
        m, P = self._x0
        for r in range(self.rank):
            m, P = e.elem_apply(0, elems[r], m, P)
        e.set_x0(m, P)
        return True

    def logpdf(self, y):
        if self.world == 1 and self.engine is None:
            return L.logpdf(self.model, y)
        if self._device_resident():
            try:
                out = self._logpdf_device(y)
                self._dx_ok = True
                return out
            except _lib.TGPError:
                raise
            except (RuntimeError, TypeError, NotImplementedError) as ex:
                # The collective layer refused the device-resident exchange on its FIRST use (every rank takes the same
                # branch): keep computing on the GPU, exchange the elements through the host instead.
                if getattr(self, "_dx_ok", False):
                    raise
                import warnings
                warnings.warn(f"device-resident exchange unavailable ({ex!r}); using the host transport")
                self._dx_disabled = True
        if not hasattr(self, "_x0"):
            self._x0 = self.engine.x0()
        reuse = self._forward_exchange(y)
        return self._all_reduce_sum(self.engine.logpdf(y, reuse))

    def logpdf_and_posterior_marginals(self, y, R_new, out=None):
        """(logpdf of the WHOLE series, this rank's slice of the posterior marginals) from one forward filter + RTS smoother
        (tgp_logpdf_and_posterior_marginals): the per-segment log-likelihood shares are by-products of the smoother's
        forward pass and are summed across ranks with one scalar all-reduce. out = (mean, var) of an earlier call: written in
        place (single GPU: lets the library replay the call from its recorded hipGraph)."""
        if self.world == 1 and self.engine is None:
            return L.logpdf_and_posterior_marginals(self.model, y, R_new, out=out)
        if self._device_resident():
            import torch
            mean, var = self._posterior_marginals_device(y, R_new)
            st = self.engine.stream()
            with torch.cuda.stream(st):          # the W shares of the log marginal likelihood: one more 4-double collective on the same transport
                stats = self._buffers()[2]
                stats.zero_()
                stats[0] = self.engine.last_segment_lml
                self.comm.all_reduce_sum(stats)
                total = float(stats.cpu()[0])
            return total, mean, var
        # host transport (gloo groups, test engines): the two calls, sharing nothing
        return (self.logpdf(y),) + tuple(self.posterior_marginals(y, R_new))

    def posterior_marginals(self, y, R_new, out=None):
        """This rank's slice of marginals(posterior(fx, y)(x)); R_new is the slice's new noise (or a scalar).  out = (mean, var) of an earlier call:
        written in place (single GPU)."""
        if self.world == 1 and self.engine is None:
            return L.posterior_marginals(self.model, y, R_new, out=out)
        if self._device_resident():
            try:
                out = self._posterior_marginals_device(y, R_new)
                self._dx_ok = True
                return out
            except _lib.TGPError:
                raise
            except (RuntimeError, TypeError, NotImplementedError) as ex:
                if getattr(self, "_dx_ok", False):
                    raise
                import warnings
                warnings.warn(f"device-resident exchange unavailable ({ex!r}); using the host transport")
                self._dx_disabled = True
        if not hasattr(self, "_x0"):
            self._x0 = self.engine.x0()
        e = self.engine
        reuse = self._forward_exchange(y)
        rev, xfm, xfP, _ = e.smoother_forward(y, reuse)
        d = e.d
        packed = np.concatenate([rev, xfm, np.asarray(xfP).reshape(-1)])
        allp = self._all_gather(packed)
        nrev = len(rev)
        if self.rank == self.world - 1:
            xs = None
        else:
            last = allp[self.world - 1]
            m, P = last[nrev:nrev + d], last[nrev + d:].reshape(d, d)
            for r in range(self.world - 1, self.rank, -1):
                m, P = e.elem_apply(1, allp[r][:nrev], m, P)
            xs = (m, P)
        y_arr = y[0] if isinstance(y, tuple) else y
        return e.smoother_backward(xs, R_new, y_arr)
