"""ctypes binding of libtgp_hip.so (include/tgp_hip.h). This is the only way the Python host mirror
reaches the device: there is NO CPU fallback -- if the HIP library is missing or no GPU is visible the
product path raises."""
import ctypes
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtgp_hip.so")

# flags / codes / options (mirror include/tgp_hip.h)
SHARED_A, SHARED_a, SHARED_Q, SHARED_H, SHARED_h, SHARED_R = (1 << i for i in range(6))
SHARED_ALL = 0x3F
SMALL_OUTPUT = 1 << 8
DEVICE_PTRS = 1 << 16
IN_DEVICE = 1 << 16
OUT_DEVICE = 1 << 17
REUSE_REDUCE = 1 << 18
OK, EINVAL, ENOTPD, EHIP, EUNSUPPORTED = 0, 1, 2, 3, 4
OPT_CHUNK, OPT_PROFILE, OPT_VARIANT, OPT_FUSE_SCAN, OPT_GROUP, OPT_TIMING, OPT_SPLIT_SMOOTHER, OPT_DENSE_STRUCTURE, OPT_GRAPH, OPT_DENSE_FUSED, OPT_SHARED_PARTS, OPT_STEADY, OPT_SDE_CLOSED_FORM = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13
OPT_SWEEP, OPT_SWEEP_CHUNK, OPT_SWEEP_WARMUP, OPT_SWEEP_WARMUP_BACK = 14, 15, 16, 17
OPT_STREAM_MIN_T = 18
OPT_WIDE = 19

_dp = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64 = ctypes.c_int64
_u32 = ctypes.c_uint32
_vp = ctypes.c_void_p

_SIGS = {
    "tgp_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int]),
    "tgp_destroy": (ctypes.c_int, [_vp]),
    "tgp_last_error": (ctypes.c_char_p, [_vp]),
    "tgp_set_option": (ctypes.c_int, [_vp, ctypes.c_int, _i64]),
    "tgp_set_stream": (ctypes.c_int, [_vp, _vp]),
    "tgp_get_stream": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "tgp_stream_synchronize": (ctypes.c_int, [_vp]),
    "tgp_version": (ctypes.c_char_p, []),
    "tgp_bind_host_thread": (ctypes.c_int, [ctypes.c_int]),
    "tgp_kernel_variant": (ctypes.c_int, [_vp]),
    "tgp_graph_replays": (_i64, [_vp]),
    "tgp_steady_steps": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "tgp_sweep_info": (ctypes.c_int, [_vp, _vp, _vp]),
    "tgp_model_set": (ctypes.c_int, [_vp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u32] + [_vp] * 8),
    "tgp_model_set_sde": (ctypes.c_int, [_vp, _i64, ctypes.c_int, ctypes.c_int, _u32] + [_vp] * 10),
    "tgp_model_set_x0": (ctypes.c_int, [_vp, _vp, _vp]),
    "tgp_logpdf": (ctypes.c_int, [_vp, _vp, _vp, _u32, _dp]),
    "tgp_logpdf_grad": (ctypes.c_int, [_vp, _vp, _vp, _u32, ctypes.c_int] + [_vp] * 8 + [_dp, _vp]),
    "tgp_logpdf_grad_sde": (ctypes.c_int, [_vp, _vp, _vp, _u32, ctypes.c_int] + [_vp] * 10 + [ctypes.c_double, _dp, _vp]),
    "tgp_logpdf_adjoint": (ctypes.c_int, [_vp, _vp, _u32, _dp] + [_vp] * 8),
    "tgp_adjoint_record_size": (ctypes.c_int, [ctypes.c_int]),
    "tgp_steady_plan": (ctypes.c_int, [ctypes.c_int] + [_vp] * 8 + [_i64, _vp, _vp, _vp, _vp]),
    "tgp_wide_plan": (ctypes.c_int, [ctypes.c_int] + [_vp] * 8 + [_i64, ctypes.c_int, _vp, _vp, _vp, _vp]),
    "tgp_segment_plan": (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "tgp_segment_logpdf_and_posterior_marginals": (ctypes.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _dp]),
    "tgp_adjoint_finish": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _i64] + [_vp] * 8),
    "tgp_filter": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _dp]),
    "tgp_posterior": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "tgp_posterior_marginals": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _dp]),
    "tgp_logpdf_and_posterior_marginals": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _dp, _vp, _vp]),
    "tgp_posterior_marginals_at": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _u32, _vp, _vp, _dp]),
    "tgp_marginals": (ctypes.c_int, [_vp, _u32, _vp, _vp]),
    "tgp_rand": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    "tgp_posterior_rand": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "tgp_logpdf_noise": (ctypes.c_int, [_vp, _vp, _u32, ctypes.c_double, _vp]),
    "tgp_pair_statistic": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "tgp_elem_size": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "tgp_segment_reduce": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp]),
    "tgp_elem_apply": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "tgp_elem_combine": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "tgp_smoother_forward": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _dp]),
    "tgp_smoother_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "tgp_shard_slot_size": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "tgp_shard_reduce": (ctypes.c_int, [_vp, _vp, _vp, _u32, _vp]),
    "tgp_shard_fold": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "tgp_shard_logpdf": (ctypes.c_int, [_vp, _vp]),
    "tgp_shard_smoother_forward": (ctypes.c_int, [_vp, _vp]),
    "tgp_shard_smoother_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _u32, _vp, _vp, _dp]),
    "tgp_shard_steady_slot_size": (ctypes.c_int, [ctypes.c_int]),
    "tgp_shard_steady_begin": (ctypes.c_int, [_vp, _vp, _u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "tgp_shard_steady_finish": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _u32, _vp, _vp, _dp, ctypes.POINTER(ctypes.c_int)]),
    "tgp_create_multi": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "tgp_destroy_multi": (ctypes.c_int, [_vp]),
    "tgp_multi_last_error": (ctypes.c_char_p, [_vp]),
    "tgp_multi_ndev": (ctypes.c_int, [_vp]),
    "tgp_multi_transport": (ctypes.c_char_p, [_vp]),
    "tgp_multi_handle": (_vp, [_vp, ctypes.c_int]),
    "tgp_multi_segment": (ctypes.c_int, [_i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "tgp_multi_set_option": (ctypes.c_int, [_vp, ctypes.c_int, _i64]),
    "tgp_multi_model_set": (ctypes.c_int, [_vp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u32] + [_vp] * 8),
    "tgp_multi_logpdf": (ctypes.c_int, [_vp, _vp, _vp, _u32, _dp]),
    "tgp_multi_posterior_marginals": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "tgp_multi_logpdf_and_posterior_marginals": (ctypes.c_int, [_vp, _vp, _vp, _vp, _u32, _dp, _vp, _vp]),
    "tgp_last_timing": (ctypes.c_int, [_vp, _dp, _dp, _dp]),
    "tgp_profile_reset": (ctypes.c_int, [_vp]),
    "tgp_profile_count": (ctypes.c_int, [_vp]),
    "tgp_profile_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, _dp, ctypes.POINTER(_i64)]),
    "tgp_profile_empty_launch": (ctypes.c_int, [_vp]),
}
EXPORTS = tuple(_SIGS)

_LIB = None


class TGPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtgp_hip error {code}: {msg}")
        self.code = code


class NotPositiveDefinite(TGPError):
    """Mirrors Julia's PosDefException / DomainError on this path (lgc.jl:135,250; lgssm.jl:235)."""


class Unsupported(TGPError, NotImplementedError):
    """TGP_EUNSUPPORTED: a combination the device engine does not implement (Julia: MethodError-class)."""


def load():
    """dlopen the in-tree HIP library; raises (never falls back) if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH) and shutil.which("hipcc") and os.environ.get("TGP_NO_AUTOBUILD") != "1":
            # a fresh checkout on a box with the ROCm toolchain: build the product in-tree (~3 minutes, once)
            subprocess.call(["make", "-s", "-j", str(os.cpu_count() or 4), "-C", os.path.join(_HERE, "csrc")])
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C temporalgps.jl_amd/csrc). There is no CPU fallback.")
        # PyTorch ships its own copy of the HIP runtime (same soname). If libtgp_hip.so pulled in the system
        # copy first, a later torch.cuda initialisation in the same process finds "No HIP GPUs"; so let
        # torch's runtime load and initialise first whenever torch is present (torch is only plumbing here:
        # device tensors, streams, torch.distributed).
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def bind_host_thread(device=0):
    """Bind the calling thread to the CPUs next to `device` (tgp_bind_host_thread); True if it was done."""
    return load().tgp_bind_host_thread(int(device)) == OK


def ptr(x):
    """Address of a NumPy array / torch tensor / int device pointer / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(type(x))


def is_device(x):
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


class Handle:
    """Owns one tgp_handle (one HIP stream on one device)."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.tgp_create(ctypes.byref(h), int(device))
        if rc != OK:
            raise TGPError(rc, "tgp_create failed (is a GPU visible?)")
        self.h = h
        self.device = device
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.tgp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == OK:
            return
        msg = self.lib.tgp_last_error(self.h).decode()
        if rc == ENOTPD:
            raise NotPositiveDefinite(rc, msg)
        if rc == EINVAL and "Dimension mismatch" in msg:
            raise ValueError(msg)
        if rc == EUNSUPPORTED:
            raise Unsupported(rc, msg)
        raise TGPError(rc, msg)

    def set_option(self, opt, value):
        self.check(self.lib.tgp_set_option(self.h, opt, int(value)))

    def last_timing(self):
        """needs set_option(OPT_TIMING, 1) before the call that is being timed"""
        k, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self.lib.tgp_last_timing(self.h, ctypes.byref(k), ctypes.byref(a), ctypes.byref(b))
        return dict(kernel_ms=k.value, h2d_ms=a.value, d2h_ms=b.value)

    def sweep_info(self):
        """diagnostics of the sweep engine (TGP_OPT_SWEEP) for the last logpdf / posterior-marginals call"""
        import numpy as np
        info, dist = np.zeros(8, dtype=np.int64), np.zeros(2)
        self.check(self.lib.tgp_sweep_info(self.h, info.ctypes.data, dist.ctypes.data))
        keys = ("served", "C", "W", "Wb", "waves", "attempts", "status", "state")
        out = {k: int(v) for k, v in zip(keys, info)}
        out["dist_f"], out["dist_b"] = float(dist[0]), float(dist[1])
        return out

    def profile_reset(self):
        self.lib.tgp_profile_reset(self.h)

    def profile(self):
        out = {}
        buf = ctypes.create_string_buffer(128)
        for i in range(self.lib.tgp_profile_count(self.h)):
            ms, calls = ctypes.c_double(), _i64()
            self.lib.tgp_profile_get(self.h, i, buf, 128, ctypes.byref(ms), ctypes.byref(calls))
            out[buf.value.decode()] = dict(total_ms=ms.value, calls=calls.value)
        return out
