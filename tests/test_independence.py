"""IndependenceSampler (tinyDA/proposal.py:65-129): oracle, host class and device path against a trace produced by the
reference (tests/golden/g12_independence.npz)."""
import numpy as np
import pytest
import scipy.stats as st

from oracle import tinyda_oracle as orc


def _level(g):
    return orc.LinearGaussianLevel(g["A"], g["data"], "iso", float(g["noise_var"]), orc.MVNPrior(g["prior_mean"], g["prior_cov"]))


def test_oracle_replays_reference_chain(golden):
    g = golden("g12_independence")
    res = orc.run_mh(_level(g), dict(kind="indep", q_mean=g["q_mean"], q_cov=g["q_cov"]), g["theta0"], g["z"], g["u"])
    assert np.array_equal(res["accepted"][:, 1:], g["accepted"][:, 1:])
    np.testing.assert_allclose(res["logpost"], g["logpost"], rtol=1e-12)
    np.testing.assert_allclose(res["theta"], g["theta"], rtol=1e-12, atol=1e-14)


def test_host_class_follows_reference_acceptance(golden):
    import tinyda_amd as tda

    g = golden("g12_independence")
    q = st.multivariate_normal(g["q_mean"], g["q_cov"])
    prop = tda.IndependenceSampler(q)
    post = tda.Posterior(st.multivariate_normal(g["prior_mean"], g["prior_cov"]),
                         tda.GaussianLogLike(g["data"], float(g["noise_var"]) * np.eye(len(g["data"]))), tda.LinearModel(g["A"]))
    # step 1 of chain 0 of the reference trace: proposal = q_mean + chol(q_cov) z
    x = post.create_link(g["theta"][0, 0])
    y = post.create_link(g["q_mean"] + np.linalg.cholesky(g["q_cov"]) @ g["z"][0, 0])
    alpha = prop.get_acceptance(y, x)
    assert (g["u"][0, 0] < alpha) == bool(g["accepted"][0, 1])
    low = prop._lowering()
    assert low["kind"] == 4 and np.allclose(low["q_mean"], g["q_mean"]) and np.allclose(low["C_"], g["q_cov"])


@pytest.mark.gpu
def test_device_replays_reference_chain(golden):
    from tinyda_amd.engine import Engine

    g = golden("g12_independence")
    N, T1, d = g["theta"].shape
    e = Engine(N, d, seed=1)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
    e.set_proposal(4, g["q_cov"], q_mean=g["q_mean"])
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    params, stats, acc = e.run_host(T1 - 1)
    e.close()
    assert np.array_equal(acc, np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(params, np.swapaxes(g["theta"][:, 1:], 0, 1), rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("d,m,N,block", [(64, 256, 40, 0), (7, 20, 21, 33)])
def test_device_forward_mode_vs_oracle(d, m, N, block):
    """Philox mode (k_rng on the second stream computes -|z|^2/2) incl. split runs and a checkpoint."""
    from tinyda_amd.engine import Engine

    rng = np.random.default_rng(d)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = rng.standard_normal(d)
    y = A @ truth + 0.2 * rng.standard_normal(m)
    cov_post = np.linalg.inv(A.T @ A / 0.04 + np.eye(d))
    mean_post = cov_post @ (A.T @ y / 0.04)
    q_mean, q_cov = mean_post + 0.02 * rng.standard_normal(d), 1.5 * cov_post
    theta0 = mean_post + 0.1 * rng.standard_normal((N, d))
    T = 150
    e = Engine(N, d, seed=31, chain_offset=5, block_steps=block)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.04)
    e.set_proposal(4, q_cov, q_mean=q_mean)
    e.init(theta0)
    z, u = e.set_export(T)
    p1, s1, a1 = e.run_host(60)
    blob = e.get_state()
    p2, s2, a2 = e.run_host(T - 60)
    e.set_state(blob)
    e.close()
    stats, acc = np.concatenate([s1, s2]), np.concatenate([a1, a2])
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.04, orc.MVNPrior(np.zeros(d), np.eye(d)))
    ref = orc.run_mh(lvl, dict(kind="indep", q_mean=q_mean, q_cov=q_cov), theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    assert 0.02 < acc.mean() < 0.98


@pytest.mark.gpu
def test_sample_api_independence(golden):
    import tinyda_amd as tda

    g = golden("g12_independence")
    post = tda.Posterior(st.multivariate_normal(g["prior_mean"], g["prior_cov"]),
                         tda.GaussianLogLike(g["data"], float(g["noise_var"]) * np.eye(len(g["data"]))), tda.LinearModel(g["A"]))
    res = tda.sample(post, tda.IndependenceSampler(st.multivariate_normal(g["q_mean"], g["q_cov"])), 200, n_chains=8, seed=4)
    assert res["sampler"] == "MH" and res.get("backend", "hip") != "host"
    link = res["chain_5"][-1]
    assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)
