"""CPU tier: the C-ABI library loads and exports every symbol include/tgp_hip.h declares (no compute)."""
import os
import re

import pytest

from tests._util import ROOT


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "temporalgps.jl_amd", "libtgp_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    import temporalgps_jl_amd as tgp
    lib = tgp._lib.load()
    header = open(os.path.join(ROOT, "include", "tgp_hip.h")).read()
    declared = set(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tgp_hip.h but not exported"
    assert declared == set(tgp._lib.EXPORTS), declared ^ set(tgp._lib.EXPORTS)


def test_option_and_flag_constants_of_the_binding_match_the_header():
    """tgp_set_option takes bare integers: the ctypes binding's OPT_* / flag constants must be the header's #defines."""
    import temporalgps_jl_amd as tgp
    header = open(os.path.join(ROOT, "include", "tgp_hip.h")).read()
    defs = {}
    for name, val in re.findall(r"#define\s+(TGP_[A-Za-z0-9_]+)\s+(\(?[0-9a-fx]+u?\s*(?:<<\s*[0-9]+)?\)?)", header):
        defs[name] = eval(val.replace("u", ""))
    opts = {k: v for k, v in vars(tgp._lib).items() if k.startswith("OPT_")}
    assert len(opts) >= 12
    for k, v in opts.items():
        assert defs.get("TGP_" + k) == v, (k, v, defs.get("TGP_" + k))
    for k in ("SHARED_A", "SHARED_a", "SHARED_Q", "SHARED_H", "SHARED_h", "SHARED_R"):
        assert defs["TGP_" + k] == getattr(tgp._lib, k), k
    assert len(set(opts.values())) == len(opts)          # no two options share a number


def test_host_side_monoid_ops_match_definition():
    """tgp_elem_apply / tgp_elem_combine are pure host functions of the ABI (no GPU needed)."""
    import ctypes
    import numpy as np
    import temporalgps_jl_amd as tgp
    lib = tgp._lib.load()
    d = 3
    n = lib.tgp_elem_size(1, d)
    assert n == d * d + d + d * (d + 1) // 2 and lib.tgp_elem_size(0, d) == d * d + 2 * d + d * (d + 1)
    rng = np.random.default_rng(0)

    def pack_affine(E, g, L):
        return np.concatenate([E.T.reshape(-1), g, np.array([L[i, j] for j in range(d) for i in range(j + 1)])])

    def sym(n_):
        X = rng.standard_normal((n_, n_))
        return X @ X.T
    E1, g1, L1, E2, g2, L2 = rng.standard_normal((d, d)), rng.standard_normal(d), sym(d), rng.standard_normal((d, d)), rng.standard_normal(d), sym(d)
    out = np.zeros(n)
    p = lambda a: a.ctypes.data
    e1, e2 = pack_affine(E1, g1, L1), pack_affine(E2, g2, L2)
    assert lib.tgp_elem_combine(1, d, p(e1), p(e2), p(out)) == 0
    want = pack_affine(E2 @ E1, E2 @ g1 + g2, E2 @ L1 @ E2.T + L2)
    np.testing.assert_allclose(out, want, rtol=1e-12, atol=1e-12)
    m, P = rng.standard_normal(d), sym(d)
    mo, Po = np.zeros(d), np.zeros((d, d))
    Pc = np.ascontiguousarray(P.T)
    assert lib.tgp_elem_apply(1, d, p(e1), p(m), p(Pc), p(mo), p(Po)) == 0
    np.testing.assert_allclose(mo, E1 @ m + g1, rtol=1e-12)
    np.testing.assert_allclose(Po.T, E1 @ P @ E1.T + L1, rtol=1e-12)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import temporalgps_jl_amd as tgp
    with pytest.raises(tgp._lib.TGPError):
        tgp._lib.Handle(0)


def test_every_ccall_of_the_julia_glue_names_a_declared_symbol_with_the_declared_number_of_arguments():
    """julia/TemporalGPsHIP.jl cannot run here (no Julia in the image): at least hold each `ccall((:sym, libtgp), Ret, (types...), args...)`
    against include/tgp_hip.h -- the symbol is declared and the ccall's type tuple has as many entries as the C prototype has parameters."""
    header = open(os.path.join(ROOT, "include", "tgp_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {}
    for name, args in re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = args.strip()
        protos[name] = 0 if args in ("", "void") else args.count(",") + 1
    src = open(os.path.join(ROOT, "julia", "TemporalGPsHIP.jl")).read()
    calls = list(re.finditer(r"ccall\(\(:(tgp_[a-z0-9_]+), libtgp\),\s*[A-Za-z]+,\s*\(", src))
    assert len(calls) >= 15
    for m in calls:
        name = m.group(1)
        assert name in protos, f"{name}: ccall'ed by the Julia glue, not declared in include/tgp_hip.h"
        depth, i, n, seen = 1, m.end(), 0, False          # walk the type tuple "(T1, T2, ...)"
        while depth:
            ch = src[i]
            if ch in "({[":
                depth += 1
            elif ch in ")}]":
                depth -= 1
            elif ch == "," and depth == 1:
                n += 1
            elif not ch.isspace():
                seen = True
            i += 1
        body = src[m.end():i - 1].strip()
        ntypes = 0 if not body else n + (0 if body.endswith(",") else 1)
        assert seen or ntypes == 0
        assert ntypes == protos[name], f"{name}: the ccall passes {ntypes} argument types, the header declares {protos[name]}"
