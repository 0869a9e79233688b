import sys
import numpy as np
sys.path.insert(0, ".")
import temporalgps_jl_amd as tgp
from oracle import lgssm_ref as ref
from tests import _util as U
from tests.test_gpu_parity import to_device_model
from tests.test_golden import load_case
for name in ("random_tv_d6", "random_tv_d4"):
    model, g = load_case(name)
    y = g("y")
    lp = float(g("logpdf"))
    fm, fP = ref.filter_(model, y)
    for chunk in (0, 2, 4, 8, 9, 16, 33):
        dm = to_device_model(tgp, model)
        dm.handle().set_option(tgp._lib.OPT_CHUNK, chunk)
        got = tgp.logpdf(dm, y)
        m, P = tgp._filter(dm, y)
        bad = np.nonzero(np.abs(m - fm).max(axis=1) > 1e-7)[0]
        hs = U.hostsim_run(model, 5, y=y, L0=chunk or 8, BS=256, want_elem=True)["elem"]
        el = np.empty(tgp._lib.load().tgp_elem_size(0, len(model["x0m"])))
        hd = dm.handle()
        hd.check(hd.lib.tgp_segment_reduce(hd.h, y.ctypes.data, None, 0, el.ctypes.data))
        print(f"{name} chunk={chunk}: dlogpdf={got - lp:.3e} bad steps={bad[:6]} n_bad={len(bad)} max|elem-host|={np.abs(el - hs).max():.2e}")
