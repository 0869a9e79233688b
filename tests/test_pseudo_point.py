"""the pseudo-point (DTC / VFE) path of space_time/pseudo_point.jl.
CPU tier: the oracle's literal state-space restatement (oracle/components.py dtc_components / build_lgssm_dtc, kind
'bottleneck') against the dense sparse-GP formulas (oracle/dense_gp.py), i.e. the reference's own test
(test/space_time/pseudo_point.jl:92-100, rtol 1e-6) on its kernel list; and the product's host-side component
construction against the oracle's.
GPU tier: dtc / elbo / approx_posterior_marginals through the device LGSSM against both."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import dense_gp as dg
from oracle import lgssm_ref as ref

SEP1 = (1.0, ("se",), ("matern12",))
SEP2 = (1.0, ("se",), ("matern52",))
KERNELS = {
    "separable-1": [SEP1],
    "separable-2": [SEP2],
    "scaled-separable": [(0.5, ("matern52",), ("matern32",))],
    "stretched-separable": [(1.0, ("se",), ("stretched", 1.3, ("matern12",)))],
    "sum-separable-1": [SEP1, SEP2],
    "sum-separable-2": [(1.3,) + SEP1[1:], (0.95,) + SEP2[1:]],
}


def _case(seed=0, N=2, M=2, T=3, dt=0.3):
    rng = np.random.default_rng(seed)
    r, z = rng.standard_normal(N), rng.standard_normal(M)
    t = ("regular", 0.0, dt, T)
    return rng, r, z, t, oc.times(t)


@pytest.mark.parametrize("name", list(KERNELS))
def test_oracle_statespace_dtc_elbo_equal_dense(name):
    terms = KERNELS[name]
    rng, r, z, t, tt = _case()
    x, zz = dg.grid_points(r, tt), dg.grid_points(z, tt)
    y = rng.standard_normal(len(x[0]))
    noise = np.full(len(y), 0.1)
    d1, d2 = oc.dtc_statespace(terms, z, r, t, 0.1, y), dg.dtc_dense(terms, x, zz, noise, y)
    assert abs(d1 - d2) <= 1e-6 * abs(d2)
    e1, e2 = oc.elbo_statespace(terms, z, r, t, 0.1, y), dg.elbo_dense(terms, x, zz, noise, y)
    assert abs(e1 - e2) <= 1e-6 * abs(e2)


def _product_kernel(terms):
    from temporalgps_jl_amd import lti_sde as S
    from temporalgps_jl_amd import space_time as ST

    def one(spec):
        name = spec[0]
        if name == "se":
            return ST.SEKernel()
        if name == "stretched":
            return one(spec[2]).stretch(spec[1])
        return {"matern12": S.Matern12Kernel, "matern32": S.Matern32Kernel, "matern52": S.Matern52Kernel}[name]()
    ks = [(s, ST.Separable(one(a), one(b))) for s, a, b in terms]
    out = None
    for s, k in ks:
        kk = k if s == 1.0 else s * k
        out = kk if out is None else out + kk
    return out


@pytest.mark.parametrize("name", list(KERNELS))
def test_product_host_components_equal_oracle(name):
    """host-side construction only (no device): A, a, Q, the projection, the fan-out and x0"""
    from temporalgps_jl_amd import lti_sde as S
    from temporalgps_jl_amd import pseudo_point as pp
    from temporalgps_jl_amd import space_time as ST
    terms = KERNELS[name]
    rng, r, z, t, tt = _case(seed=3, N=4, M=3, T=5)
    grid = ST.RectilinearGrid(r, S.RegularSpacing(0.0, 0.3, 5))
    A, a, Q, (Ct, Hb, hb), (m0, P0) = pp.lgssm_components(pp.dtcify(z, _product_kernel(terms)), grid)
    oA, oa, oQ, oHb, ohb, oCt, om, oP = oc.dtc_components(terms, z, r, t)
    for got, want in ((A, oA), (a, oa), (Q, oQ), (Hb, oHb), (hb, ohb), (Ct, oCt), (m0, om), (P0, oP)):
        np.testing.assert_allclose(np.broadcast_to(got, np.broadcast_shapes(got.shape, want.shape)),
                                   np.broadcast_to(want, np.broadcast_shapes(got.shape, want.shape)), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(pp.kernel_diagonals(pp.dtcify(z, _product_kernel(terms)), grid), oc.dtc_kernel_diagonals(terms, r, t),
                               rtol=1e-12)


# ------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(KERNELS))
def test_gpu_dtc_elbo_posterior_equal_dense(name):
    """test/space_time/pseudo_point.jl:92-111 with the device backend; tolerances as there (1e-6 / 1e-7)."""
    from temporalgps_jl_amd import lti_sde as S
    from temporalgps_jl_amd import pseudo_point as pp
    from temporalgps_jl_amd import space_time as ST
    terms = KERNELS[name]
    rng, r, z, t, tt = _case(seed=11)
    k = _product_kernel(terms)
    grid = ST.RectilinearGrid(r, S.RegularSpacing(0.0, 0.3, 3))
    x, zz = dg.grid_points(r, tt), dg.grid_points(z, tt)
    y = rng.standard_normal(len(x[0]))
    noise = np.full(len(y), 0.1)
    d_dense = dg.dtc_dense(terms, x, zz, noise, y)
    assert abs(pp.dtc(k, grid, 0.1, y, z) - d_dense) <= 1e-6 * abs(d_dense)
    e_dense = dg.elbo_dense(terms, x, zz, noise, y)
    assert abs(pp.elbo(k, grid, 0.1, y, z) - e_dense) <= 1e-6 * abs(e_dense)
    x_pr = rng.standard_normal(10)
    xs = dg.grid_points(x_pr, tt)
    pm, pv = dg.vfe_posterior_marginals(terms, x, zz, noise, y, xs)
    gm, gv = pp.approx_posterior_marginals(k, grid, 0.1, y, z, x_pr)
    np.testing.assert_allclose(gm.reshape(-1), pm, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.sqrt(gv.reshape(-1)), np.sqrt(pv), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_gpu_dtc_long_series_with_missing_against_oracle_statespace():
    """larger case (M = 4 pseudo-points, Matern-3/2 in time: state dimension 8, 20 observations per step, T = 300, 25 %
    missing) against the oracle's literal BottleneckLGC recursion."""
    from temporalgps_jl_amd import lti_sde as S
    from temporalgps_jl_amd import pseudo_point as pp
    from temporalgps_jl_amd import space_time as ST
    terms = [(0.8, ("se",), ("matern32",))]
    rng = np.random.default_rng(2)
    N, M, T = 20, 4, 300
    r, z = rng.standard_normal(N), np.linspace(-1.5, 1.5, M)
    t = ("regular", 0.0, 0.1, T)
    y = rng.standard_normal(T * N)
    miss = rng.random(T * N) < 0.25
    k = _product_kernel(terms)
    grid = ST.RectilinearGrid(r, S.RegularSpacing(0.0, 0.1, T))
    want = oc.dtc_statespace(terms, z, r, t, 0.2, y, missing=miss)
    ym = y.copy()
    ym[miss] = np.nan
    got = pp.dtc(k, grid, 0.2, ym, z)
    assert abs(got - want) <= 1e-6 * abs(want)
    want_e = oc.elbo_statespace(terms, z, r, t, 0.2, y)
    assert abs(pp.elbo(k, grid, 0.2, y, z) - want_e) <= 1e-6 * abs(want_e)


@pytest.mark.gpu
def test_gpu_regular_in_time_with_ragged_slices_equals_the_grid_with_those_points_missing():
    """RegularInTime (regular_in_time.jl:8-89) with a different number of points, at different places, per time slice: the slices are
    padded to the longest one and the padding marked missing. Identity the reference's tests rest on (missing == marginalised,
    test/models/missings.jl:94-115): dtc, elbo and the approximate posterior marginals of the ragged data set equal those of the
    full rectilinear grid whose absent points are missing -- the oracle's literal BottleneckLGC recursion on that grid."""
    from temporalgps_jl_amd import lti_sde as S
    from temporalgps_jl_amd import pseudo_point as pp
    from temporalgps_jl_amd import space_time as ST
    terms = [(0.8, ("se",), ("matern32",)), (0.3, ("se",), ("matern12",))]
    rng = np.random.default_rng(12)
    N, M, T = 9, 3, 80
    r, z = np.sort(rng.standard_normal(N)), np.linspace(-1.2, 1.2, M)
    t = ("regular", 0.0, 0.15, T)
    keep = rng.random((T, N)) < 0.7
    keep[5] = False                       # a time slice with no observation at all
    keep[6, :] = True
    y_full = rng.standard_normal((T, N))
    sig_full = 0.1 + 0.2 * rng.random((T, N))
    k = _product_kernel(terms)
    times = S.RegularSpacing(0.0, 0.15, T)
    ragged = ST.RegularInTime(times, [r[keep[i]] for i in range(T)])
    assert len(ragged) == keep.sum() and ragged.shape2 == (T, N)
    y_r, sig_r = y_full[keep], sig_full[keep]              # flat, slice after slice
    want = oc.dtc_statespace(terms, z, r, t, sig_full.reshape(-1), y_full.reshape(-1), missing=(~keep).reshape(-1))
    got = pp.dtc(k, ragged, sig_r, y_r, z)
    assert abs(got - want) <= 1e-8 * abs(want)
    # the same through the rectilinear-grid code path with NaN == missing (the padding changes nothing but the slots' positions)
    grid = ST.RectilinearGrid(r, times)
    ym = np.where(keep, y_full, np.nan)
    assert abs(pp.dtc(k, grid, sig_full, ym, z) - want) <= 1e-8 * abs(want)
    e_grid = pp.elbo(k, grid, sig_full, ym, z)
    e_ragged = pp.elbo(k, ragged, sig_r, y_r, z)
    assert abs(e_ragged - e_grid) <= 1e-8 * abs(e_grid)
    x_pr = np.array([-0.9, 0.05, 0.7, 1.4])
    m_g, v_g = pp.approx_posterior_marginals(k, grid, sig_full, ym, z, x_pr)
    m_r, v_r = pp.approx_posterior_marginals(k, ragged, sig_r, y_r, z, x_pr)
    np.testing.assert_allclose(m_r, m_g, rtol=0, atol=1e-8)
    np.testing.assert_allclose(v_r, v_g, rtol=0, atol=1e-8)
    # flat <-> time form round trip (regular_in_time.jl:53-65)
    np.testing.assert_array_equal(ragged.unpad(ragged.pad(y_r, np.nan)), y_r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["separable-2", "sum-separable-2"])
def test_gpu_elbo_with_one_space_point(name):
    """N = 1: the device returns the prior marginals of a scalar-observation model as (T,), and the trace term subtracted a (T,) from
    a (T, 1) array (broadcast to (T, T)) until round 3 (scripts/stress_pseudo_point.py)."""
    from temporalgps_jl_amd import lti_sde as S
    from temporalgps_jl_amd import pseudo_point as pp
    from temporalgps_jl_amd import space_time as ST
    terms = KERNELS[name]
    rng, r, z, t, tt = _case(seed=5, N=1, M=3, T=9)
    k = _product_kernel(terms)
    grid = ST.RectilinearGrid(r, S.RegularSpacing(0.0, 0.3, 9))
    x, zz = dg.grid_points(r, tt), dg.grid_points(z, tt)
    y = rng.standard_normal(len(x[0]))
    noise = np.full(len(y), 0.1)
    d_dense, e_dense = dg.dtc_dense(terms, x, zz, noise, y), dg.elbo_dense(terms, x, zz, noise, y)
    assert abs(pp.dtc(k, grid, 0.1, y, z) - d_dense) <= 1e-6 * abs(d_dense)
    assert abs(pp.elbo(k, grid, 0.1, y, z) - e_dense) <= 1e-6 * abs(e_dense)
