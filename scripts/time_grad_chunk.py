import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde as P, lgssm as L, _lib
T = 10_000_000
for base in ("matern52", "matern32"):
    fx = P.to_sde(P.GP(P.to_kernel((base,))))(P.RegularSpacing(0.0, 0.1, T), 0.1)
    model = fx.build_lgssm()
    d = model.dim
    y = torch.randn(T, dtype=torch.float64, device="cuda:0")
    tang = [dict(A=0.01 * np.eye(d), Q=0.01 * np.eye(d)), dict(R=1.0), dict(H=np.ones(d) * 0.1)]
    hd = model.handle()
    for chunk in (0, 77, 153):
        hd.set_option(_lib.OPT_CHUNK, chunk)
        for _ in range(2): L.logpdf_and_grad(model, y, tang)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); L.logpdf_and_grad(model, y, tang); ts.append(time.perf_counter() - t0)
        print(f"RESULT {base} d={d} chunk={chunk}: {min(ts)*1e3:.3f} ms for logpdf + 3 tangents")
