#!/usr/bin/env python3
"""Where the chunks of a series switch to the mean-only steps (TGP_STEADY_DEBUG histogram of tgp_api.hip). Usage: steady_hist.py [T] [chunk]"""
import os
import sys

import numpy as np

os.environ["TGP_STEADY_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import temporalgps_jl_amd as tgp  # noqa: E402
from temporalgps_jl_amd import _lib, lti_sde  # noqa: E402

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for spec, dt in ((("matern52",), 0.1), (("matern52",), 0.01), (("matern32",), 0.1), (("sum", ("matern52",), ("matern12",)), 0.1)):
    m = lti_sde.build_lgssm(lti_sde.to_kernel(spec), lti_sde.RegularSpacing(0.0, dt, T), 0.1, device=0)
    if chunk:
        m.handle().set_option(_lib.OPT_CHUNK, chunk)
    y = torch.as_tensor(np.random.default_rng(1).standard_normal(T), device="cuda:0")
    print(spec, dt, file=sys.stderr, flush=True)
    tgp.logpdf_and_posterior_marginals(m, y, np.array([1e-18]))
