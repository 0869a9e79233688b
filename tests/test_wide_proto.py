"""CPU tier: the ALGORITHM of the wide-state engine (scripts/wide_proto.py: csrc/tgp_wide.hip's host plan and kernel structure restated in NumPy) against
the oracle -- logpdf against the literal restatement of lgssm.jl:147-165, posterior marginals against the dense GP on the model's own covariance
function.  The HIP kernels: tests/test_gpu_wide.py; the product's own host plan: tests/test_wide_plan.py."""
import importlib.util
import os

import numpy as np
import pytest
from scipy.linalg import cho_factor, cho_solve, toeplitz

from oracle import components as oc
from oracle import lgssm_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {
    9: ("product", ("matern52",), ("stretched", 0.7, ("matern52",))),
    12: ("product", ("matern32",), ("approx_periodic", 3, 1.0)),
    20: ("product", ("approx_periodic", 5, 1.3), ("matern32",)),
}


@pytest.fixture(scope="module")
def proto():
    spec = importlib.util.spec_from_file_location("wide_proto", os.path.join(ROOT, "scripts", "wide_proto.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("d", sorted(KERNELS))
@pytest.mark.parametrize("chunks", (1, 5))
def test_chunked_stationary_recursions_equal_the_oracle(proto, d, chunks):
    T = 1600
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.2, T), 0.1)
    rng = np.random.default_rng(d)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    pl = proto.plan(model, T)
    assert pl is not None and pl["n0"] + pl["n1"] < T
    Rn = rng.random(T) * 0.2 + 0.01
    lml, mean, var = proto.run(pl, y, Rn, chunks=chunks)
    lp_ref = ref.logpdf(model, y)
    assert abs(lml - lp_ref) <= 1e-10 * abs(lp_ref), (lml, lp_ref)
    A, H, P, R = model["A"][0], model["H"][0], model["x0P"], float(model["R"][0])
    c, v = np.empty(T), P @ H
    for k in range(T):
        c[k] = H @ v
        v = A @ v
    K = toeplitz(c)
    cf = cho_factor(K + R * np.eye(T), lower=True)
    m_gp = K @ cho_solve(cf, y)
    v_gp = np.diag(K) - np.einsum("ij,ji->i", K, cho_solve(cf, K)) + Rn
    assert np.max(np.abs(mean - m_gp)) <= 1e-8 * max(1.0, np.abs(m_gp).max()), np.max(np.abs(mean - m_gp))
    assert np.max(np.abs(var - v_gp)) <= 1e-8 * max(1.0, v_gp.max()), np.max(np.abs(var - v_gp))
