"""Chunk-length sweep of the combined call (logpdf + posterior marginals) and of logpdf alone, LTI layout, T = 1e7."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib, lti_sde as P
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for name, k in (("d=3", P.Matern52Kernel()), ("d=2", P.Matern32Kernel()), ("d=4", P.Matern52Kernel() + P.Matern12Kernel())):
    fx = P.to_sde(P.GP(k))(P.RegularSpacing(0.0, 0.1, T), 0.1)
    y = torch.randn(T, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
    Rn = torch.full((1,), 0.1, dtype=torch.float64, device="cuda:0")
    row = []
    for chunk in [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,48,64,77,96,128,153,192,256".split(","))]:
        model = fx.build_lgssm() if chunk == 0 else P.build_lgssm(fx.f.f.kernel, fx.x, fx.sigma2, fx.f.f.mean, 0)
        model.handle_options[_lib.OPT_SHARED_PARTS] = 0
        if chunk: model.handle_options[_lib.OPT_CHUNK] = chunk
        out = None
        for _ in range(3):
            res = tgp.logpdf_and_posterior_marginals(model, y, Rn, out=out); out = res[1:]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tgp.logpdf_and_posterior_marginals(model, y, Rn, out=out)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20): tgp.logpdf(model, y)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        row.append(f"{chunk}: {(t1 - t0) / 20 * 1e3:.3f}/{(t2 - t1) / 20 * 1e3:.3f}")
        del model
    print(name, "chunk: combined ms / logpdf ms |", "  ".join(row), flush=True)
