"""GPU tier: the STREAMING kernels of the stationary-gain engine (round 6) -- csrc/tgp_lml.hip (logpdf alone: y read once, no halo, the runs closed
by quadratic forms) and csrc/tgp_post.hip (logpdf + posterior marginals: persistent waves, one run of 1024-step tiles each, outputs one
tile late) -- against the oracle's sequential restatement of lgssm.jl:99-238 (oracle/seq_kalman.c), through the C ABI.
Tolerances as everywhere: logpdf 1e-10 relative, marginals 1e-8.  Lengths sit around every tile / run / workgroup boundary of both kernels."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import seq_kalman as sk

pytestmark = pytest.mark.gpu

KERNELS = {
    1: ("matern12",),
    2: ("matern32",),
    3: ("matern52",),
    4: ("sum", ("matern52",), ("matern12",)),
    5: ("sum", ("matern52",), ("matern32",)),
    6: ("sum", ("matern52",), ("stretched", 0.4, ("matern52",))),
    7: ("sum", ("matern52",), ("stretched", 0.5, ("matern32",)), ("scaled", 0.3, ("matern32",))),
    8: ("sum", ("matern52",), ("stretched", 2.0, ("matern52",)), ("stretched", 0.5, ("matern32",))),
}


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def device_model(tgp, model, min_T=0):
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])
    # (by default the streaming kernels serve from their measured crossovers on -- 5e6 / 3e6 steps; here: every length)
    dm.handle_options[tgp._lib.OPT_STREAM_MIN_T] = min_T
    return dm


def kernels_of(tgp, dm, fn):
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    out = fn()
    names = set(hd.profile())
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    return out, names


def draw(model, seed):
    T, d = model["T"], len(model["x0m"])
    rng = np.random.default_rng(seed)
    return sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))


# tiles of 1024 (d <= 4) or 2048 (d >= 5) steps behind a head of 16..144 steps: one run with a partial tile, exactly one tile, one step more, several
# runs, a partial last tile, more tiles than wave slots (4096 / 2048 runs: the runs hold several tiles)
LML_LENGTHS = (70, 700, 2047 + 64, 2048 + 64, 2049 + 64, 5000, 16_384 + 65, 300_001, 2048 * 2048 + 64 + 17)


@pytest.mark.parametrize("d", sorted(KERNELS))
def test_streaming_logpdf(tgp, d):
    for T in LML_LENGTHS if d in (3, 6) else LML_LENGTHS[::2]:
        for dt, s2 in ((0.1, 0.1), (0.01, 1e-3)) if T < 100_000 else ((0.1, 0.1),):
            model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), s2)
            y = draw(model, 7 * d + T % 13)
            dm = device_model(tgp, model)
            lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
            ref = sk.logpdf(model, y)
            assert abs(lp - ref) <= 1e-10 * abs(ref), (d, T, dt, lp, ref)
            # (a series shorter than the head of a slowly settling model is not the stationary-gain engines': the plan declines, the general engine serves it)
            if T >= 700 and dt == 0.1:
                assert len(names) == 1 and next(iter(names)).startswith("k_lml_stream"), (d, T, names)


def test_streaming_logpdf_device_pointer_off_the_16_byte_boundary(tgp):
    import torch
    for d in (3, 6):
        T = 300_001
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, 5)
        buf = torch.zeros(T + 1, dtype=torch.float64, device="cuda")
        buf[1:] = torch.from_numpy(y).cuda()
        dm = device_model(tgp, model)
        lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, buf[1:]))
        ref = sk.logpdf(model, y)
        assert abs(lp - ref) <= 1e-10 * abs(ref)
        assert next(iter(names)).startswith("k_lml_stream"), names


# tiles of 1024 steps behind the head; run 0 holds the first tile, each of the last three tiles is a run of its own, the runs between them hold
# C tiles each (C = 1 up to 2044 tiles, 2 beyond ...): one tile, two, four (no middle run), five, a partial last tile, C = 2
POST_LENGTHS = (1100, 1024 + 64, 2048 + 64 - 1, 2048 + 64 + 1, 4096 + 64, 5000, 5 * 1024 + 64 + 3, 100_000, 300_001, 2048 * 1024 + 64 + 1025)


@pytest.mark.parametrize("d", (1, 2, 3))
def test_streaming_posterior(tgp, d):
    rng = np.random.default_rng(d)
    for T in POST_LENGTHS if d == 3 else POST_LENGTHS[::2]:
        for dt, s2 in ((0.1, 0.1), (0.01, 1e-3)) if T < 50_000 else ((0.1, 0.1),):
            model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, dt, T), s2)
            y = draw(model, 3 * d + T % 11)
            for per_step in (False, True):
                Rn = rng.random(T) + 0.05 if per_step else np.array([0.3])
                m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
                lp_ref = sk.logpdf(model, y)
                dm = device_model(tgp, model)
                (lp, mean, var), names = kernels_of(tgp, dm, lambda: tgp.logpdf_and_posterior_marginals(dm, y, Rn))
                assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (d, T, lp, lp_ref)
                assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8, (d, T, per_step)
                # (series shorter than head + tail tables run the whole plan in front of the launch: k_steady_one with the head inside the kernel)
                if T >= 5000:
                    assert names == {"k_post_stream"}, (d, T, names)


def test_streaming_kernels_serve_from_their_crossovers_on(tgp):
    """TGP_OPT_STREAM_MIN_T = -1 (the default): k_steady_one below 5e6 / 3e6 steps, the streaming kernels from there on; results agree across the switch"""
    Rn = np.array([0.2])
    for T, want_lml, want_post in ((1_000_000, False, False), (4_000_000, False, True), (6_000_000, True, True)):
        model = oc.build_lgssm(KERNELS[3], ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, T % 7)
        dm = device_model(tgp, model, min_T=-1)
        lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, y))
        assert next(iter(names)).startswith("k_lml_stream") == want_lml, (T, names)
        (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
        assert (names == {"k_post_stream"}) == want_post, (T, names)
        ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
        assert abs(lp - ref) <= 1e-10 * abs(ref)
        assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8


def test_streaming_posterior_repeated_calls_and_odd_pointers(tgp):
    """the runs' exchange records carry the call's sequence number: a second call on the same handle must not read the first call's; outputs off
    the 16-byte boundary go to k_steady_one"""
    import torch
    T = 300_001
    model = oc.build_lgssm(KERNELS[3], ("regular", 0.0, 0.1, T), 0.1)
    dm = device_model(tgp, model)
    Rn = np.array([0.2])
    for seed in (1, 2, 3):
        y = draw(model, seed)
        m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
        (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, y, Rn))
        assert names == {"k_post_stream"}, names
        assert np.max(np.abs(mean - m_ref)) <= 1e-8 and np.max(np.abs(var - v_ref)) <= 1e-8
    y = draw(model, 9)
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    yd = torch.from_numpy(y).cuda()
    buf_m, buf_v = torch.zeros(T + 1, dtype=torch.float64, device="cuda"), torch.zeros(T + 1, dtype=torch.float64, device="cuda")
    (mean, var), names = kernels_of(tgp, dm, lambda: tgp.posterior_marginals(dm, yd, torch.from_numpy(Rn).cuda(), out=(buf_m[1:], buf_v[1:])))
    assert all(n.startswith("k_steady_one") for n in names), names
    assert np.max(np.abs(mean.cpu().numpy() - m_ref)) <= 1e-8 and np.max(np.abs(var.cpu().numpy() - v_ref)) <= 1e-8


def test_plan_kept_between_calls_is_dropped_when_another_model_plans(tgp):
    """the core of a handle's last plan is kept while model and length stand (logpdf, then posterior, of one model plan once); the stages behind the core
    read a per-thread workspace, so a plan of ANOTHER model of the same state dimension in between must make the first handle plan again -- and a
    changed noise variance on the same handle is a new model"""
    T = 300_001
    Rn = np.array([0.25])
    models = [oc.build_lgssm(("matern52",), ("regular", 0.0, dt, T), s2) for dt, s2 in ((0.1, 0.1), (0.05, 0.3))]
    ys = [draw(m, 11 + i) for i, m in enumerate(models)]
    refs = [(sk.logpdf(m, y),) + tuple(sk.posterior_marginals(m, y, Rn)) for m, y in zip(models, ys)]
    for min_T in (0, -1):      # the streaming kernels / k_steady_one with the head on the host
        dms = [device_model(tgp, m, min_T=min_T) for m in models]
        for order in ((0, 1, 0, 1), (0, 0, 1, 1), (1, 0, 0, 1)):
            for i in order:
                lp = tgp.logpdf(dms[i], ys[i])
                assert abs(lp - refs[i][0]) <= 1e-10 * abs(refs[i][0]), (min_T, order, i)
            for i in order:
                mean, var = tgp.posterior_marginals(dms[i], ys[i], Rn)
                assert np.max(np.abs(mean - refs[i][1])) <= 1e-8 and np.max(np.abs(var - refs[i][2])) <= 1e-8, (min_T, order, i)
                lp = tgp.logpdf(dms[i], ys[i])
                assert abs(lp - refs[i][0]) <= 1e-10 * abs(refs[i][0]), (min_T, order, i)


@pytest.mark.parametrize("d", (3, 5))
def test_reverse_ordered_lti_prior_runs_as_the_forward_twin_on_the_flipped_series(tgp, d):
    """lgssm.jl:87-91, 161-165: a Reverse-ordered model with shared blocks is the Forward model on reverse(y) -- logpdf and _filter go through one flip
    pass and the one-launch kernels (TGP_REVERSE_FLIP=0: the general engine as before); against the literal restatement with ordering = 'R'"""
    import torch
    from oracle import lgssm_ref as ref
    for T in (900, 6000):
        model = dict(oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1), ordering="R")
        rng = np.random.default_rng(T + d)
        y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        lp_ref = ref.logpdf(model, y)
        fm_ref, fP_ref = ref.filter_(model, y)
        tr = tgp.GaussMarkovModel(tgp.Reverse, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
        dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
        for yy in (y, torch.from_numpy(y).cuda()):
            lp, names = kernels_of(tgp, dm, lambda: tgp.logpdf(dm, yy))
            assert abs(lp - lp_ref) <= 1e-10 * abs(lp_ref), (d, T, lp, lp_ref)
            assert any(n.startswith("k_steady_one") or n.startswith("k_lml_stream") for n in names), names
            (fm, fP), names = kernels_of(tgp, dm, lambda: tgp._filter(dm, yy))
            fm, fP = (fm.cpu().numpy(), fP.cpu().numpy()) if hasattr(fm, "cpu") else (fm, fP)
            assert np.max(np.abs(fm - np.asarray(fm_ref))) <= 1e-8 and np.max(np.abs(fP - np.asarray(fP_ref))) <= 1e-8, (d, T)
            if d <= 6:
                assert "k_flip_rows" in names and any(n.startswith("k_filter_one") for n in names), names
