#!/usr/bin/env python3
"""A/B of the stationary-covariance steps (TGP_OPT_STEADY, passes 2 / 3 of shared-layout models with d <= 3): bit-identity of
logpdf / filter / posterior marginals with the option on and off, and the time of the combined call either way.
Usage: ab_steady.py [T]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import temporalgps_jl_amd as tgp  # noqa: E402
from temporalgps_jl_amd import _lib, lti_sde  # noqa: E402

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
SPECS = {"matern12_d1": ("matern12",), "matern32_d2": ("matern32",), "matern52_d3": ("matern52",)}
for name, spec in SPECS.items():
    for dt in (0.1, 0.01):
        m = lti_sde.build_lgssm(lti_sde.to_kernel(spec), lti_sde.RegularSpacing(0.0, dt, T), 0.1, device=0)
        rng = np.random.default_rng(1)
        y = torch.as_tensor(rng.standard_normal(T), device="cuda:0")
        Rn = np.array([1e-18])
        res = {}
        for on in (0, 1):
            m.handle().set_option(_lib.OPT_SHARED_PARTS, 0)
            m.handle().set_option(_lib.OPT_STEADY, on)
            out = (torch.empty(T, dtype=torch.float64, device="cuda:0"), torch.empty(T, dtype=torch.float64, device="cuda:0"))
            lp, mean, var = tgp.logpdf_and_posterior_marginals(m, y, Rn, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tgp.logpdf_and_posterior_marginals(m, y, Rn, out=out)
            torch.cuda.synchronize()
            t_pm = (time.perf_counter() - t0) / 10
            lp2 = tgp.logpdf(m, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tgp.logpdf(m, y)
            torch.cuda.synchronize()
            t_lp = (time.perf_counter() - t0) / 10
            res[on] = (lp, mean.clone(), var.clone(), lp2, t_pm, t_lp)
        a, b = res[0], res[1]
        same = a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
        print(f"{name} dt={dt} T={T}: {'bit-identical' if same else 'DIFFERENT'}  combined {a[4]*1e3:.3f} -> {b[4]*1e3:.3f} ms   logpdf {a[5]*1e3:.3f} -> {b[5]*1e3:.3f} ms"
              f"   (lml {a[0]!r} / {b[0]!r}, max |dmean| {float((a[1]-b[1]).abs().max()):.2e}, max |dvar| {float((a[2]-b[2]).abs().max()):.2e})", flush=True)
