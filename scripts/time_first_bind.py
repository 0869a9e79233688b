"""Cost of binding the FIRST model of a state dimension in a process: the run-time known-answer check (variant_selftest) runs then."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import temporalgps_jl_amd as tgp
from tests import _util as U
from tests.test_gpu_parity import to_device_model
t0 = time.perf_counter()
dm = to_device_model(tgp, U.random_lgssm(np.random.default_rng(0), False, 2, 1000)); dm.handle(); tgp.logpdf(dm, np.zeros(1000))
print(f"process warm-up (library load, first kernel): {time.perf_counter() - t0:.2f} s", flush=True)
for d in (3, 5, 6, 7, 8, 9, 14, 16):
    for tv in (False, True):
        model = U.random_lgssm(np.random.default_rng(d), tv, d, 2000)
        t0 = time.perf_counter()
        dm = to_device_model(tgp, model)
        dm.handle()
        t1 = time.perf_counter()
        dm2 = to_device_model(tgp, model)
        dm2.handle()
        t2 = time.perf_counter()
        print(f"d={d:2d} {'per-step' if tv else 'lti     '}: first bind {t1 - t0:6.2f} s, second bind {1e3 * (t2 - t1):6.1f} ms", flush=True)
