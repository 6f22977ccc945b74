"""Batched host-callback forward models (tda.BatchedModel / tda_engine_set_level_callback): the device engine with the
model evaluated on the host for all chains at once, against the oracle running the same NumPy model chain by chain on the
exported Philox stream."""
import numpy as np
import pytest
import scipy.stats as st

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu

M = 70  # more outputs than lanes: the wave strides twice


def np_model(theta):
    theta = np.atleast_2d(theta)
    W = 0.1 + 0.01 * ((np.arange(M)[:, None] * 7 + np.arange(theta.shape[1])[None, :] * 3) % 11)
    return np.tanh(theta @ W.T) + 0.25 * theta[:, [0]] * theta[:, [-1]]


@pytest.mark.parametrize("kind", ["am", "pcn", "grw_diag"])
def test_callback_model_matches_oracle(kind):
    from tinyda_amd.engine import Engine

    d, N, T = 6, 21, 130
    rng = np.random.default_rng(14)
    truth = 0.5 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(M)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    pm, pv = (np.zeros(d), np.ones(d)) if kind == "pcn" else (0.1 * np.arange(d), 0.5 + 0.1 * np.arange(d))
    calls = []

    def fn(thetas):
        calls.append(thetas.shape)
        return np_model(thetas)

    e = Engine(N, d, seed=91, chain_offset=3, block_steps=33)
    e.set_prior(pm, np.diag(pv))
    noise = 0.05 ** 2 * (1.0 + 0.1 * np.arange(M))
    if kind == "grw_diag":
        e.set_level_callback(0, fn, y, 1, noise)
        lvl = orc.CallableGaussianLevel(np_model, y, "diag", noise, orc.MVNPrior(pm, np.diag(pv)))
        e.set_proposal(0, 2e-3 * np.eye(d), scaling=1.0, adaptive=True, period=20)
        prop = dict(kind="grw", C=2e-3 * np.eye(d), scaling=1.0, adaptive=True, period=20)
    else:
        e.set_level_callback(0, fn, y, 0, [0.05 ** 2])
        lvl = orc.CallableGaussianLevel(np_model, y, "iso", 0.05 ** 2, orc.MVNPrior(pm, np.diag(pv)))
        if kind == "am":
            e.set_proposal(2, 2e-3 * np.eye(d), t0=40, period=20)
            prop = dict(kind="am", C0=2e-3 * np.eye(d), t0=40, period=20)
        else:
            e.set_proposal(1, None, scaling=0.03)
            prop = dict(kind="pcn", scaling=0.03)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    e.close()
    assert calls == [(N, d)] * (T + 1)  # one call for the initial links, one per step
    ref = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(params, np.swapaxes(ref["theta"][:, 1:], 0, 1), rtol=1e-9, atol=1e-11)  # AM: device Cholesky vs LAPACK
    assert 0.05 < acc.mean() < 0.95


@pytest.mark.parametrize("inplace", [False, True])
def test_callback_model_through_sample_api(inplace):
    import tinyda_amd as tda

    d = 6
    rng = np.random.default_rng(2)
    truth = 0.3 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(M)
    post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(M)),
                         tda.BatchedModel(np_model, M) if not inplace else
                         tda.BatchedModel(lambda t, out: np.copyto(out, np_model(t)), M, inplace=True))
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d), t0=50, period=50), 300, n_chains=8,
                     initial_parameters=truth, seed=3)
    assert res["sampler"] == "MH" and res["backend"] == "hip" and res["n_chains"] == 8
    link = res["chain_3"][-1]
    assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)
    assert 0.0 < np.mean(res["chain_3"].accepted[1:]) < 1.0


def test_callback_exception_and_bad_shape_surface_in_python():
    from tinyda_amd.engine import Engine

    d, N = 3, 5
    state = {"n": 0}

    def flaky(thetas):
        state["n"] += 1
        if state["n"] == 4:
            raise RuntimeError("solver diverged")
        return thetas[:, :2].copy()

    e = Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_callback(0, flaky, np.zeros(2), 0, [1.0])
    e.set_proposal(0, 0.1 * np.eye(d))
    e.init(np.zeros((N, d)))
    with pytest.raises(RuntimeError, match="solver diverged"):
        e.run_host(10)
    e.close()

    e = Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_callback(0, lambda t: np.zeros((N, 3)), np.zeros(2), 0, [1.0])
    e.set_proposal(0, 0.1 * np.eye(d))
    with pytest.raises(ValueError, match="returned shape"):
        e.init(np.zeros((N, d)))
    e.close()


def test_callback_nonfinite_outputs_are_rejections():
    """NaN model outputs make the proposal's posterior NaN -> alpha = 0 (proposal.py:253-258)."""
    from tinyda_amd.engine import Engine

    d, N, T = 3, 8, 40
    first = {"done": False}

    def fn(thetas):
        out = thetas[:, :2].copy()
        if first["done"]:
            out[::2] = np.nan  # every even chain's proposal fails from the first step on
        first["done"] = True
        return out

    e = Engine(N, d, seed=5)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_callback(0, fn, np.zeros(2), 0, [1.0])
    e.set_proposal(0, 0.05 * np.eye(d))
    theta0 = 0.1 * np.ones((N, d))
    e.init(theta0)
    params, stats, acc = e.run_host(T)
    e.close()
    assert not acc[:, ::2].any() and acc[:, 1::2].any()
    assert np.array_equal(params[:, ::2], np.broadcast_to(theta0[::2], (T,) + theta0[::2].shape))
    assert np.isfinite(stats).all()
