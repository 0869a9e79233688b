"""BASELINE config 1 (Matern-3/2, d = 2, T = 1e4): time per combined call with and without hipGraph replay."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib, lti_sde as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
for kern, name in ((P.Matern32Kernel(), "matern32 d=2"), (P.Matern52Kernel(), "matern52 d=3")):
    for graph, chunk in ((0, 0), (1, 0), (0, 4), (0, 8), (0, 16), (1, 8), (0, 32), (0, 64)):
        fx = P.to_sde(P.GP(kern))(P.RegularSpacing(0.0, 0.1, T), 0.1)
        model = fx.build_lgssm()
        model.handle_options[_lib.OPT_GRAPH] = graph
        if chunk:
            model.handle_options[_lib.OPT_CHUNK] = chunk
        y = torch.randn(T, dtype=torch.float64, device="cuda:0")
        Rn = torch.full((1,), 0.1, dtype=torch.float64, device="cuda:0")
        out = None
        for _ in range(5):
            res = tgp.logpdf_and_posterior_marginals(model, y, Rn, out=out)
            out = res[1:]
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            res = tgp.logpdf_and_posterior_marginals(model, y, Rn, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            tgp.logpdf(model, y)
        dl = (time.perf_counter() - t0) / n
        hd = model.handle()
        print(f"{name} T={T} graph={graph} chunk={chunk}: combined {dt * 1e6:.1f} us/call ({T / dt:.3e} steps/s), logpdf {dl * 1e6:.1f} us/call, "
              f"replays {hd.lib.tgp_graph_replays(hd.h)}", flush=True)
