"""Golden vectors (tests/golden/lgssm_golden.npz, made by tests/golden/make_golden.py):
   CPU tier  -- the oracle (NumPy + C restatements) and the CPU emulation of the engine reproduce them;
   GPU tier  -- the HIP path, through the C ABI, reproduces them (tolerances as in test_gpu_parity.py)."""
import os

import numpy as np
import pytest

from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk
from tests import _util as U

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lgssm_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in G.files})


def load_case(name):
    g = lambda k: G[f"{name}/{k}"]
    model = dict(ordering="F" if int(g("ordering")) == 0 else "R", kind="scalar", T=len(g("y")), A=g("A"), a=g("a"), Q=g("Q"),
                 H=g("H"), h=g("h"), R=g("R"), x0m=g("x0m"), x0P=g("x0P"))
    return model, g


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    model, g = load_case(name)
    y = g("y")
    assert ref.logpdf(model, y) == pytest.approx(float(g("logpdf")), rel=1e-13)
    np.testing.assert_allclose(ref.rand(model, g("eps_t"), g("eps_e"), g("eps_0")), y, rtol=1e-13, atol=1e-13)
    assert ref.logpdf_missing(model, y, g("missing")) == pytest.approx(float(g("logpdf_missing")), rel=1e-13)
    if model["ordering"] == "F" and len(model["x0m"]) <= 8:
        assert sk.logpdf(model, y) == pytest.approx(float(g("logpdf")), rel=1e-12)
        m, v = sk.posterior_marginals(model, y, g("Rnew"))
        np.testing.assert_allclose(m, g("post_mean"), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(v, g("post_var"), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name", [n for n in NAMES if f"{n}/logpdf_mp" in G.files])
def test_oracle_rounding_error_against_extended_precision(name):
    """The fp64 oracle against the SAME recursions in 50-digit arithmetic (oracle/lgssm_mp.py, frozen in the golden file): its
    own rounding error is orders of magnitude below every parity tolerance stated against it (1e-10 logpdf, 1e-8 posterior)."""
    model, g = load_case(name)
    y = g("y")
    lp = ref.logpdf(model, y)
    assert abs(lp - float(g("logpdf_mp"))) <= 1e-12 * max(1.0, abs(lp))
    lpm = ref.logpdf_missing(model, y, g("missing"))
    assert abs(lpm - float(g("logpdf_missing_mp"))) <= 1e-12 * max(1.0, abs(lpm))
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior(model, y), g("Rnew")))
    scale = max(1.0, np.abs(g("post_mean_mp")).max())
    assert np.abs(pm - g("post_mean_mp")).max() <= 1e-11 * scale
    assert np.abs(pv - g("post_var_mp")).max() <= 1e-11 * max(1.0, np.abs(g("post_var_mp")).max())
    if len(model["x0m"]) <= 8:
        m, v = sk.posterior_marginals(model, y, g("Rnew"))
        assert np.abs(m - g("post_mean_mp")).max() <= 1e-10 * scale


@pytest.mark.parametrize("name", NAMES)
def test_engine_emulation_reproduces_golden(name):
    model, g = load_case(name)
    y = g("y")
    r = U.hostsim_run(model, 0, y=y, L0=5, BS=3)
    assert r["lml"] == pytest.approx(float(g("logpdf")), rel=1e-10)
    r = U.hostsim_run(model, 0, y=y, missing=g("missing"), L0=5, BS=3)
    assert r["lml"] == pytest.approx(float(g("logpdf_missing")), rel=1e-10)
    if model["ordering"] == "F":
        r = U.hostsim_run(model, 2, y=y, L0=5, BS=3, Rnew=g("Rnew"))
        np.testing.assert_allclose(r["mean"], g("post_mean"), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(r["var"], g("post_var"), rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(name):
    import temporalgps_jl_amd as tgp
    from tests.test_gpu_parity import to_device_model
    model, g = load_case(name)
    y = g("y")
    dm = to_device_model(tgp, model)
    assert tgp.logpdf(dm, y) == pytest.approx(float(g("logpdf")), rel=1e-10)
    ym = y.copy()
    ym[g("missing")] = np.nan
    assert tgp.logpdf(dm, ym) == pytest.approx(float(g("logpdf_missing")), rel=1e-10)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, g("filter_m"), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, g("filter_P"), rtol=1e-8, atol=1e-9)
    mm, mv = tgp.marginals(dm)
    np.testing.assert_allclose(mm, g("marg_mean"), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(mv, g("marg_var"), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(tgp.rand((g("eps_t"), g("eps_e"), g("eps_0")), dm), y, rtol=1e-9, atol=1e-9)
    if model["ordering"] == "F":
        post = tgp.posterior(dm, y)
        np.testing.assert_allclose(post.transitions.As, g("post_G"), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(post.transitions.as_, g("post_g"), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(post.transitions.Qs, g("post_L"), rtol=1e-8, atol=1e-9)
        pm, pv = tgp.posterior_marginals(dm, y, g("Rnew"))
        np.testing.assert_allclose(pm, g("post_mean"), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(pv, g("post_var"), rtol=1e-8, atol=1e-9)
        # ... and against the 50-digit evaluation of the same recursions (not the fp64 oracle)
        assert tgp.logpdf(dm, y) == pytest.approx(float(g("logpdf_mp")), rel=1e-10)
        np.testing.assert_allclose(pm, g("post_mean_mp"), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(pv, g("post_var_mp"), rtol=1e-8, atol=1e-9)
        pm, pv = tgp.posterior_marginals(dm, ym, g("Rnew"))
        np.testing.assert_allclose(pm, g("post_mean_missing"), rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(pv, g("post_var_missing"), rtol=1e-8, atol=1e-9)
