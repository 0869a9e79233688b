"""TEST / BASELINE INFRASTRUCTURE (never on the product path): the ALL-CORE CPU leg of the measurement.

The reference's scan is sequential (src/util/scan.jl:15-28), so its own CPU number is a one-core number (oracle/seq_kalman.c).
To keep the GPU speed-up from being flattered, this module times the time-PARALLEL formulation on every host core: the
product's own chunk functions and scan monoids (temporalgps.jl_amd/csrc/tgp_chunk.hpp, tgp_math.hpp -- the headers the HIP
kernels instantiate) compiled for the host with -O3 -march=x86-64-v3 -fopenmp by the host driver tests/hostsim/hostsim.cpp,
one chunk per thread at a time. LTI scalar-output models, d <= 8."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_ROOT, "tests", "hostsim", "hostsim.cpp")
_SO = os.path.join(_HERE, "libomp_scan.so")
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64


def build(force=False):
    deps = [_SRC] + [os.path.join(_ROOT, "temporalgps.jl_amd", "csrc", f) for f in ("tgp_math.hpp", "tgp_math_body.inc", "tgp_chunk.hpp", "tgp_chunk_body.inc")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(p) > os.path.getmtime(_SO) for p in deps):
        subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", _SO, _SRC])
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(x):
    return None if x is None else x.ctypes.data_as(_dp)


def run(model, y, what, Rnew=None, chunks_per_thread=8):
    """what: 0 logpdf -> lml; 2 posterior marginals -> (lml, mean, var). `model`: oracle dict, scalar kind, every block shared."""
    d, T = len(model["x0m"]), int(model["T"])
    assert model["kind"] == "scalar" and all(np.atleast_1d(model[k]).shape[0] == 1 for k in ("A", "a", "Q", "H", "h", "R"))
    nthr = os.cpu_count() or 1
    L0 = max(8, T // (nthr * chunks_per_thread))
    A = np.ascontiguousarray(model["A"][0].T).reshape(-1)
    Q = np.ascontiguousarray(model["Q"][0].T).reshape(-1)
    a, H = np.ascontiguousarray(model["a"][0]), np.ascontiguousarray(model["H"][0])
    h, R = np.atleast_1d(np.asarray(model["h"], dtype=np.float64)), np.atleast_1d(np.asarray(model["R"], dtype=np.float64))
    x0m, x0P = np.ascontiguousarray(model["x0m"], dtype=np.float64), np.ascontiguousarray(model["x0P"].T, dtype=np.float64).reshape(-1)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    lml = ctypes.c_double()
    mean = var = xfm = xfP = Rn = None
    if what == 2:
        mean, var, xfm, xfP = np.zeros(T), np.zeros(T), np.zeros(d), np.zeros((d, d))
        Rn = np.ascontiguousarray(np.atleast_1d(Rnew), dtype=np.float64)
    rc = lib().hostsim_run(d, 1, 0, 1, what, int(L0), 4096, _i64(T), 0, _p(A), _i64(0), _p(a), _i64(0), _p(Q), _i64(0), _p(H), _i64(0), _p(h), _i64(0),
                           _p(R), _i64(0), _p(yv), None, _p(x0m), _p(x0P), ctypes.byref(lml), None, None, None, None, None, _p(xfm), _p(xfP),
                           _p(Rn), _i64(0), _p(mean), _p(var), None, None, None, None, None, None)
    assert rc == 0, rc
    return lml.value if what == 0 else (lml.value, mean, var)


def logpdf(model, y):
    return run(model, y, 0)


def posterior_marginals(model, y, Rnew):
    return run(model, y, 2, Rnew)
