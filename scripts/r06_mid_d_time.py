import time, torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib as L, lti_sde as P
L.bind_host_thread(0)
for spec, name in ((("approx_periodic", 7, 1.0), "approx_periodic d=14"), (("sum", ("matern52",), ("matern52",), ("matern32",), ("matern32",)), "sum d=10"), (("product", ("matern52",), ("matern52",)), "prod 52x52 d=9"), (("product", ("matern32",), ("approx_periodic", 3, 1.0)), "d=12")):
    for T in (1_000_000, 10_000_000):
        try:
            model = P.build_lgssm(P.to_kernel(spec), P.RegularSpacing(0.0, 0.1, T), 0.1)
            y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
            Rn = torch.full((1,), 0.1, dtype=torch.float64, device="cuda:0")
            tgp.logpdf(model, y); torch.cuda.synchronize()
            t0 = time.perf_counter(); 
            for _ in range(3): tgp.logpdf(model, y)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            tgp.posterior_marginals(model, y, Rn); torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(3): tgp.posterior_marginals(model, y, Rn)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            hd = model.handle(); hd.set_option(L.OPT_PROFILE, 1); hd.profile_reset(); tgp.logpdf(model, y); names = list(hd.profile()); hd.set_option(L.OPT_PROFILE, 0)
            print(f"{name} d={model.dim} T={T}: logpdf {(t1-t0)/3*1e3:.3f} ms, posterior marginals {(t3-t2)/3*1e3:.3f} ms  {names[:4]}")
        except Exception as ex:
            print(name, T, "ERR", repr(ex)[:200])
