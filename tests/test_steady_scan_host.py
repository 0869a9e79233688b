"""CPU tier: the ALGORITHM of the stationary-gain scan engine (csrc/tgp_steady.hip) restated in NumPy with the kernels' structure --
setup tables (head gains, stationary gains, tail / head variances, powers, tile couplings incl. the ragged last tile), 512-step tiles
of 64 lanes x 8 steps with constant-coefficient wave scans, tile carries -- against the oracle's sequential restatement of
lgssm.jl:99-238.  scripts/steady_proto.py is the development prototype the HIP kernels were written from; the HIP path itself is
checked in tests/test_gpu_steady_scan.py."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def proto():
    spec = importlib.util.spec_from_file_location("steady_proto", os.path.join(ROOT, "scripts", "steady_proto.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("kern,dt,T", [(("matern52",), 0.1, 2100), (("matern32",), 0.1, 1541), (("sum", ("matern52",), ("matern32",)), 0.1, 1800),
                                       (("matern52",), 0.03, 2600), (("matern52",), 0.1, 1027)])
def test_prototype_equals_sequential_recursion(proto, kern, dt, T):
    model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), 0.1)
    d = len(model["x0m"])
    rng = np.random.default_rng(T)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    out = proto.run(model, y, 0.05)
    assert out is not None
    lml, mean, var, tab = out
    lp_ref = ref.logpdf(model, y)
    post = ref.posterior(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, np.array([0.05])))
    assert abs(lml - lp_ref) <= 1e-11 * abs(lp_ref)
    np.testing.assert_allclose(mean, pm, rtol=0, atol=1e-9)
    np.testing.assert_allclose(var, pv, rtol=1e-9, atol=1e-11)
    assert 0 < tab["n0"] < 512 and tab["n1"] > 0


def test_prototype_declines_short_series(proto):
    """shorter than head + tail: the path does not apply (the product re-runs such a call on the general engine)"""
    model = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, 540), 0.1)
    assert proto.run(model, np.zeros(540), 0.05) is None
