"""CPU tier: the host plan of the stationary-gain engine's one-launch path (csrc/tgp_steady_plan.hpp through the pure host function
tgp_steady_plan of libtgp_hip.so) and the ALGORITHM of its kernel (scripts/modal_proto.py: the kernel's structure restated in NumPy on the
product's own plan) against the oracle's sequential restatement of lgssm.jl:99-238.  The HIP kernel itself: tests/test_gpu_modal.py."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def proto():
    spec = importlib.util.spec_from_file_location("modal_proto", os.path.join(ROOT, "scripts", "modal_proto.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _draw(model, T, seed):
    d = len(model["x0m"])
    rng = np.random.default_rng(seed)
    return ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))


CASES = [
    (("matern52",), 0.1, 0.1, 9000),
    (("matern32",), 0.1, 0.1, 5541),
    (("matern12",), 0.1, 0.1, 4999),
    (("sum", ("matern52",), ("matern12",)), 0.1, 0.1, 9100),
    (("sum", ("matern52",), ("matern32",)), 0.1, 0.1, 9800),
    (("sum", ("matern52",), ("stretched", 0.4, ("matern52",))), 0.1, 0.1, 12000),
    (("sum", ("matern52",), ("stretched", 0.5, ("matern32",)), ("scaled", 0.3, ("matern32",))), 0.1, 0.3, 9000),
    (("matern52",), 0.03, 0.5, 9000),
    (("matern52",), 1.0, 1e-3, 4100),
    (("sum", ("matern52",), ("matern32",)), 0.03, 0.5, 30011),      # slow mixing: 16 tiles per workgroup
    (("sum", ("matern52",), ("matern32",)), 0.02, 0.05, 24000),
]


@pytest.mark.parametrize("kern,dt,s2,T", CASES)
def test_prototype_on_the_products_plan_equals_sequential_recursion(proto, kern, dt, s2, T):
    model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), s2)
    y = _draw(model, T, T)
    out = proto.run(model, y, 0.05)
    assert out is not None, proto.plan(model, T)
    lml, mean, var, pl = out
    lp_ref = ref.logpdf(model, y)
    post = ref.posterior(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, np.array([0.05])))
    assert abs(lml - lp_ref) <= 1e-10 * abs(lp_ref), (lml, lp_ref, pl["cond_f"], pl["halo"])
    np.testing.assert_allclose(mean, pm, rtol=0, atol=1e-8)
    np.testing.assert_allclose(var, pv, rtol=0, atol=1e-8)
    assert 0 < pl["n0"] <= 623 and pl["n1"] > 0 and pl["halo"] % 16 == 0 and pl["nhs"] % 16 == 0


def test_plan_declines_what_the_one_launch_path_does_not_serve(proto):
    # a sum of two IDENTICAL kernels: the difference of the two components is unobservable, the closed loop keeps the defective
    # open-loop eigenvalue of the Matern-5/2 block -- no well-conditioned modal form (the dense kernels of tgp_steady.hip serve it)
    m = oc.build_lgssm(("sum", ("matern52",), ("matern52",)), ("regular", 0.0, 0.1, 20000), 0.1)
    assert proto.plan(m, 20000)["why"] in (1, 4, 7)
    # shorter than head + tail
    m = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.1, 100), 0.1)
    assert proto.plan(m, 100)["why"] == 3
    # mixes too slowly for the longest halo
    m = oc.build_lgssm(("matern52",), ("regular", 0.0, 0.001, 50000), 1.0)
    assert proto.plan(m, 50000)["why"] in (1, 5)


def test_modal_form_reproduces_the_closed_loop(proto):
    """the block form handed to the kernel is the stationary closed loop: fw' M^j fb == h' Phi^j (A K) for the impulse response"""
    model = oc.build_lgssm(("sum", ("matern52",), ("matern32",)), ("regular", 0.0, 0.1, 5000), 0.1)
    pl = proto.plan(model, 5000)
    assert pl["why"] == 0 and pl["npair"] >= 1
    d = pl["d"]
    A = np.asarray(model["A"]).reshape(-1, d, d)[0]
    Q = np.asarray(model["Q"]).reshape(-1, d, d)[0]
    H = np.asarray(model["H"]).reshape(-1)[:d]
    R = float(np.asarray(model["R"]).reshape(-1)[0])
    Pf = np.asarray(model["x0P"]).reshape(d, d)
    for _ in range(2000):
        Pp = A @ Pf @ A.T + Q
        S = H @ Pp @ H + R
        K = Pp @ H / S
        Pf = Pp - np.outer(K, K) * S
    Phi = A - np.outer(A @ K, H)
    P = np.array([(i ^ 1) if (i ^ 1) < d else i for i in range(d)])
    z, x = pl["fb"].copy(), A @ K
    for j in range(40):
        assert abs(pl["fw"] @ z - H @ x) <= 1e-12
        z = pl["fd"] * z + pl["fo"] * z[P]
        x = Phi @ x
