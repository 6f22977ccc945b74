"""The reference's helper and deprecated names (tinyda_amd/compat.py) on the host: no GPU."""
import warnings

import numpy as np
import pytest
from scipy import stats

import tinyda_amd as tda


def _posterior(d=3, m=8, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d))
    truth = rng.standard_normal(d)
    y = A @ truth + 0.05 * rng.standard_normal(m)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    return tda.Posterior(prior, tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(m)), tda.LinearModel(A)), A, y


def test_map_and_ml_of_a_linear_gaussian_posterior():
    post, A, y = _posterior()
    s2 = 0.05 ** 2
    H = np.eye(3) + A.T @ A / s2
    exact_map = np.linalg.solve(H, A.T @ y / s2)
    exact_ml = np.linalg.lstsq(A, y, rcond=None)[0]
    np.testing.assert_allclose(tda.get_MAP(post, initial_parameters=np.zeros(3)), exact_map, atol=1e-4)
    np.testing.assert_allclose(tda.get_ML(post, initial_parameters=np.zeros(3)), exact_ml, atol=1e-4)
    de = tda.get_MAP(post, method="differential_evolution", bounds=[(-4, 4)] * 3, seed=1, tol=1e-10)
    np.testing.assert_allclose(de, exact_map, atol=1e-3)
    assert tda.get_MAP(post).shape == (3,)  # default start: a draw from the prior


def test_gradients():
    post, A, y = _posterior()
    x = np.array([0.3, -0.2, 0.1])
    np.testing.assert_allclose(tda.grad_log_p(x, post.prior), -x)
    out = A @ x
    g = tda.grad_log_l(out, post.likelihood)
    num = np.array([(post.likelihood.loglike(out + 1e-6 * e) - post.likelihood.loglike(out - 1e-6 * e)) / 2e-6 for e in np.eye(len(out))])
    np.testing.assert_allclose(g, num, rtol=1e-5)
    jp = tda.JointPrior([stats.norm(0, 2), stats.norm(1, 0.5), stats.norm(0, 1)])
    np.testing.assert_allclose(tda.grad_log_p(x, jp), [-x[0] / 4, -(x[1] - 1) / 0.25, -x[2]], rtol=1e-4)


def test_to_xarray_layout():
    samples = {"n_chains": 2, "iterations": 5, "dimension": 3, "attribute": "parameters", "level": "fine",
               "chain_0": np.arange(15.0).reshape(5, 3), "chain_1": -np.arange(15.0).reshape(5, 3)}
    ds = tda.to_xarray(samples, ["a", "b", "c"])
    assert list(ds) == ["a", "b", "c"]
    assert np.asarray(ds["b"]).shape == (2, 5)
    np.testing.assert_array_equal(np.asarray(ds["b"])[0], samples["chain_0"][:, 1])
    np.testing.assert_array_equal(np.asarray(ds["c"])[1], samples["chain_1"][:, 2])


def test_deprecated_names_warn_and_work():
    post, A, y = _posterior()
    with pytest.warns(UserWarning, match="BlackBoxLinkFactory is deprecated"):
        bb = tda.BlackBoxLinkFactory(tda.LinearModel(A), post.prior, post.likelihood)
    x = np.array([0.1, 0.2, 0.3])
    assert bb.create_link(x).posterior == post.create_link(x).posterior

    class Mine(tda.LinkFactory):
        def evaluate_model(self, parameters):
            return A @ parameters

    with pytest.warns(UserWarning, match="LinkFactory is deprecated"):
        lf = Mine(post.prior, post.likelihood)
    np.testing.assert_allclose(lf.create_link(x).posterior, post.create_link(x).posterior)
    with pytest.warns(UserWarning, match="CompositePrior"):
        cp = tda.CompositePrior([stats.norm(), stats.uniform(0, 1)])
    assert isinstance(cp, tda.JointPrior)
    with pytest.warns(UserWarning, match="SingleDreamZ"):
        dz = tda.SingleDreamZ(M0=20, delta=1)
    assert isinstance(dz, tda.DREAMZ)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tda.Posterior(post.prior, post.likelihood, tda.LinearModel(A))


def _two_level(seed=3):
    rng = np.random.default_rng(seed)
    d, m = 3, 6
    A1 = rng.standard_normal((m, d))
    A0 = A1 + 0.05 * rng.standard_normal((m, d))
    y = A1 @ rng.standard_normal(d)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    mk = lambda A, like: tda.Posterior(prior, like(y, 0.1 * np.eye(m)), tda.LinearModel(A))
    return mk, A0, A1, d


def test_dachain_matches_the_hierarchy_driver():
    from tinyda_amd.hostloop import HierarchyChain

    mk, A0, A1, d = _two_level()
    runs = []
    for cls in ("compat", "driver"):
        posts = [mk(A0, tda.AdaptiveGaussianLogLike), mk(A1, tda.GaussianLogLike)]
        prop = tda.GaussianRandomWalk(0.05 * np.eye(d))
        np.random.seed(11)
        if cls == "compat":
            ch = tda.DAChain(posts[0], posts[1], prop, 3, initial_parameters=np.zeros(d), adaptive_error_model="state-independent")
        else:
            ch = HierarchyChain(posts, prop, [3], np.zeros(d), "state-independent")
        ch.sample(25, progressbar=False)
        runs.append(ch)
    a, b = runs
    assert len(a.chain_fine) == 26 and len(a.accepted_fine) == 26
    np.testing.assert_array_equal([ln.parameters for ln in a.chain_fine], [ln.parameters for ln in b.rungs[1].links])
    assert a.accepted_coarse == b.rungs[0].took and a.is_coarse == b.rungs[0].own
    assert len(a.chain_coarse) == len(a.is_coarse) == 1 + 25 * 4  # three coarse steps and one aligned entry per fine step
    assert len(a.promoted_coarse) == len(a.subchain_lengths) <= 25
    np.testing.assert_array_equal(a.bias.get_mu(), b.trackers[1].get_mu())


def test_mldachain_attributes():
    mk, A0, A1, d = _two_level()
    posts = [mk(A0, tda.GaussianLogLike), mk(0.5 * (A0 + A1), tda.GaussianLogLike), mk(A1, tda.GaussianLogLike)]
    np.random.seed(5)
    ch = tda.MLDAChain(posts, tda.GaussianRandomWalk(0.05 * np.eye(d)), [2, 2], initial_parameters=np.zeros(d))
    ch.sample(10, progressbar=False)
    assert len(ch.chain) == len(ch.accepted) == 11
    assert ch.posterior is posts[-1] and len(ch.levels) == 3
    assert all(hasattr(ln, "posterior") for ln in ch.chain)
    with pytest.raises(ValueError):
        tda.MLDAChain(posts, tda.GaussianRandomWalk(0.05 * np.eye(d)), [2])
