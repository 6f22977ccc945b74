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


@pytest.mark.gpu
def test_linear_hierarchy_and_dreamz_with_uniform_components():
    """The fused multi-level kernel and the DREAM(Z) kernel test the support bounds too: a 3-level MLDA run of linear levels
    against the oracle, and DREAMZ over a linear model staying inside the supports."""
    import tinyda_amd as tda
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, N, sl, n_fine, seed = 6, 19, [3, 2], 14, 77
    rng = np.random.default_rng(61)
    kinds = np.array([1, 0, 1, 0, 0, 1])
    loc = np.array([-0.5, 0.0, -0.4, 0.1, 0.0, -1.0])
    scale = np.array([1.0, 1.0, 0.9, 0.8, 1.0, 2.0])
    truth = np.array([0.42, 0.3, 0.4, -0.2, 0.1, 0.0])  # two components near the upper edge of their support
    ms = (10, 16, 24)
    As = [rng.standard_normal((m, d)) / np.sqrt(d) for m in ms]
    ys = [A @ truth + 0.2 * rng.standard_normal(len(A)) for A in As]
    theta0 = truth + 0.03 * rng.standard_normal((N, d))
    theta0[:, [0, 2]] = np.minimum(theta0[:, [0, 2]], (loc + scale)[[0, 2]] - 1e-3)
    e = Engine(N, d, seed=seed, n_levels=3)
    e.set_prior_joint(kinds, loc, scale)
    for k in range(3):
        e.set_level(k, As[k], ys[k], 0, 0.04)
    e.set_proposal(2, 5e-3 * np.eye(d), t0=20, period=10, adaptive=True, gamma=1.02)
    e.set_subchains(sl, False)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    e.close()
    us, _ = _oracle_uniforms(seed, N, rows, sl)
    prior = orc.JointPriorOracle(kinds, loc, scale)
    levels = [orc.LinearGaussianLevel(As[k], ys[k], "iso", 0.04, prior) for k in range(3)]
    res, _ = orc.run_multilevel(levels, dict(kind="am", C0=5e-3 * np.eye(d), t0=20, period=10, adaptive=True, gamma=1.02), sl, theta0,
                                np.swapaxes(z, 0, 1), us, n_fine, None)
    for k in range(3):
        sk = slice(1, None) if k == 2 else slice(None)
        assert np.array_equal(outs[k][2], res[k]["accepted"][:, sk].T), "level %d accept masks differ" % k
        np.testing.assert_allclose(outs[k][1][:, :, 2], res[k]["logpost"][:, sk].T, rtol=1e-10)
    assert (outs[0][0][:, :, 0] <= loc[0] + scale[0]).all() and (outs[0][0][:, :, 2] <= loc[2] + scale[2]).all()
    # DREAMZ over a linear model under the same kind of prior
    prior_h = tda.JointPrior([st.uniform(l, s) if k else st.norm(l, s) for k, l, s in zip(kinds, loc, scale)])
    post = tda.Posterior(prior_h, tda.GaussianLogLike(ys[2], 0.04 * np.eye(ms[2])), tda.LinearModel(As[2]))
    np.random.seed(4)
    r = tda.sample(post, tda.DREAMZ(30, delta=1), 120, n_chains=10, seed=9)
    assert r.get("backend", "hip") != "host"
    th = np.stack([np.asarray(r["chain_%d" % i].parameters) for i in range(10)])
    assert (th[:, :, 0] >= loc[0]).all() and (th[:, :, 0] <= loc[0] + scale[0]).all() and (th[:, :, 5] <= 1.0).all()
    lk = r["chain_3"][-1]
    assert np.isclose(lk.posterior, post.create_link(lk.parameters).posterior, rtol=1e-10)
