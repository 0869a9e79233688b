import time, torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib as L, lti_sde as P
L.bind_host_thread(0)
for spec, T in ((("product", ("approx_periodic", 7, 1.0), ("matern32",)), 10_000_000), (("product", ("approx_periodic", 7, 1.0), ("matern52",)), 1_000_000)):
    model = P.build_lgssm(P.to_kernel(spec), P.RegularSpacing(0.0, 0.1, T), 0.1)
    y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
    Rn = torch.full((1,), 0.1, dtype=torch.float64, device="cuda:0")
    out = (torch.empty_like(y), torch.empty_like(y))
    ts = []
    for i in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tgp.posterior_marginals(model, y, Rn, out=out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(model.dim, T, " ".join(f"{t:.2f}" for t in ts))
