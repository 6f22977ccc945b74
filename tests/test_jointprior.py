"""JointPrior (tinyDA/distributions.py:8-100): host class and oracle against values produced by the reference
(tests/golden/g10_jointprior.npz), and the device path replaying the reference's trace."""
import numpy as np
import pytest
import scipy.stats as st

from oracle import tinyda_oracle as orc


def _components(g):
    return [st.norm(l, s) if k == 0 else st.uniform(l, s) for k, l, s in zip(g["kinds"], g["loc"], g["scale"])]


def test_host_class_and_oracle_match_reference_logpdf(golden):
    import tinyda_amd as tda

    g = golden("g10_jointprior")
    prior = tda.JointPrior(_components(g))
    with np.errstate(divide="ignore"):
        host = np.array([prior.logpdf(x) for x in g["kat_points"]])
    assert np.array_equal(host, g["kat_logpdf"])  # same scipy calls in the same order
    o = orc.JointPriorOracle(g["kinds"], g["loc"], g["scale"])
    assert np.array_equal(o.logpdf(g["kat_points"]), g["kat_logpdf"])
    kinds, loc, scale = prior._lowering()
    assert np.array_equal(kinds, g["kinds"]) and np.allclose(loc, g["loc"]) and np.allclose(scale, g["scale"])
    assert prior.rvs().shape == (5,) and prior.rvs(3).shape == (3, 5)
    assert tda.JointPrior([st.norm(0, 1), st.gamma(2.0)])._lowering() is None  # not lowered: host protocol


def test_oracle_replays_reference_chain(golden):
    g = golden("g10_jointprior")
    lvl = orc.LinearGaussianLevel(g["A"], g["data"], "iso", float(g["noise_var"]), orc.JointPriorOracle(g["kinds"], g["loc"], g["scale"]))
    prop = dict(kind="grw", C=g["C"], scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]), period=int(g["period"]))
    res = orc.run_mh(lvl, prop, g["theta0"], g["z"], g["u"])
    assert np.array_equal(res["accepted"][:, 1:], g["accepted"][:, 1:])
    np.testing.assert_allclose(res["logpost"], g["logpost"], rtol=1e-12)
    np.testing.assert_allclose(res["logprior"], g["logprior"], rtol=1e-12)


@pytest.mark.gpu
def test_device_replays_reference_chain(golden):
    from tinyda_amd.engine import Engine

    g = golden("g10_jointprior")
    N, T1, d = g["theta"].shape
    e = Engine(N, d, seed=1)
    e.set_prior_joint(g["kinds"], g["loc"], g["scale"])
    e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
    e.set_proposal(0, g["C"], scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]), period=int(g["period"]))
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    _, st0 = e.current()
    params, stats, acc = e.run_host(T1 - 1)
    np.testing.assert_allclose(st0[:, 2], g["logpost"][:, 0], rtol=1e-10)
    assert np.array_equal(acc, np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(stats[:, :, 0], np.swapaxes(g["logprior"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(e.proposal_state()["scaling"], g["scaling_hist"][:, -1], rtol=1e-12)
    e.close()


@pytest.mark.gpu
def test_sample_api_with_joint_prior_stays_in_support(golden):
    import tinyda_amd as tda

    g = golden("g10_jointprior")
    prior = tda.JointPrior(_components(g))
    post = tda.Posterior(prior, tda.GaussianLogLike(g["data"], float(g["noise_var"]) * np.eye(len(g["data"]))), tda.LinearModel(g["A"]))
    res = tda.sample(post, tda.AdaptiveMetropolis(0.05 * np.eye(5), t0=50, period=25), 400, n_chains=16, seed=9)
    assert res.get("backend", "hip") != "host"
    th = np.array([l.parameters for l in res["chain_3"]])
    lo = np.where(g["kinds"] == 1, g["loc"], -np.inf)
    hi = np.where(g["kinds"] == 1, g["loc"] + g["scale"], np.inf)
    assert np.all(th >= lo) and np.all(th <= hi)
    assert np.isclose(res["chain_3"][-1].prior, prior.logpdf(th[-1]), rtol=1e-10)
