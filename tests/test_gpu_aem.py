"""GPU parity of the adaptive error model: replay of tinyDA's DAChain / MLDAChain traces with adaptive_error_model on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10
KIND = {"grw": 0, "pcn": 1, "am": 2}


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()
    return engine


def _engine(eng_mod, g, nl, sl, aem):
    from tests.test_oracle_multilevel import _prop

    th0 = g["theta0"]
    N, d = th0.shape
    m = g["A0"].shape[0]
    e = eng_mod.Engine(N, d, seed=5, n_levels=nl)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    for k in range(nl):
        if k < nl - 1:
            e.set_level(k, g["A%d" % k], g["y%d" % k], 3, float(g["noise_var"]) * np.eye(m), b=g["b%d" % k])
        else:
            e.set_level(k, g["A%d" % k], g["y%d" % k], 0, float(g["noise_var"]), b=g["b%d" % k])
    pr = _prop(g)
    if pr["kind"] == "grw":
        e.set_proposal(0, pr["C"], scaling=pr["scaling"])
    else:
        e.set_proposal(1, None, scaling=pr["scaling"])
    e.set_subchains(sl)
    e.set_error_model(aem)
    e.init(th0)
    return e


def _check(outs, g, nl, st_init):
    for k in range(nl):
        params, stats, acc = outs[k]
        ref_acc, ref_lp, ref_ll, ref_th = g["acc%d" % k], g["lp%d" % k], g["ll%d" % k], g["th%d" % k]
        if k == nl - 1:
            np.testing.assert_allclose(st_init[:, 2], ref_lp[:, 0] + ref_ll[:, 0], rtol=RTOL)
            ref_acc, ref_lp, ref_ll, ref_th = ref_acc[:, 1:], ref_lp[:, 1:], ref_ll[:, 1:], ref_th[:, 1:]
        assert np.array_equal(acc, ref_acc.T), "level %d: %d accept flips" % (k, int((acc != ref_acc.T).sum()))
        np.testing.assert_allclose(params, np.swapaxes(ref_th, 0, 1), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(stats[:, :, 0], ref_lp.T, rtol=RTOL)
        if k == nl - 1:  # coarse links are refreshed after they were recorded (update_link), the finest never is
            np.testing.assert_allclose(stats[:, :, 2], (ref_lp + ref_ll).T, rtol=RTOL)


@pytest.mark.parametrize("name", ["g8_da_aem_indep", "g8_da_aem_dep", "g8_da_aem_dep_pcn", "g8_da_aem_indep_m72", "g8_da_aem_dep_pcn_m128"])
def test_da_error_model_replay(eng_mod, golden, name):
    g = golden(name)
    L = int(g["subchain_length"])
    n_fine = g["th1"].shape[1] - 1
    e = _engine(eng_mod, g, 2, [L], str(g["aem"]))
    e.set_replay(np.swapaxes(g["z"], 0, 1), g["u0"].T)
    e.set_replay_level(1, g["u1"].T)
    _, st1 = e.level_state(1)
    outs = e.run_levels_host(n_fine)
    _check(outs, g, 2, st1)
    bias, P = e.error_model_state(0, g["A0"].shape[0])
    np.testing.assert_allclose(bias, g["bias_mu"], rtol=1e-9, atol=1e-12)
    cov = float(g["noise_var"]) * np.eye(g["A0"].shape[0]) + g["bias_sigma"]
    np.testing.assert_allclose(P, np.linalg.inv(cov), rtol=1e-7, atol=1e-9)
    e.close()


@pytest.mark.parametrize("name", ["g8_mlda_aem", "g8_mlda_aem_m100"])
def test_mlda_error_model_replay(eng_mod, golden, name):
    g = golden(name)
    nl = int(g["n_levels"])
    sl = [int(x) for x in g["subchain_lengths"]]
    n_fine = g["th%d" % (nl - 1)].shape[1] - 1
    e = _engine(eng_mod, g, nl, sl, "state-independent")
    e.set_replay(np.swapaxes(g["z"], 0, 1), g["u0"].T)
    for k in range(1, nl):
        e.set_replay_level(k, g["u%d" % k].T)
    _, stf = e.level_state(nl - 1)
    outs = e.run_levels_host(n_fine)
    _check(outs, g, nl, stf)
    e.close()


def test_sample_api_with_error_model_beyond_64_outputs():
    """tda.sample(..., adaptive_error_model=...) lowers AdaptiveGaussianLogLike levels with up to 128 outputs (two-wave
    error-model kernels); the finest-level links carry the posterior of the finest model."""
    import scipy.stats as st

    import tinyda_amd as tda

    d, m, N = 6, 100, 12
    rng = np.random.default_rng(5)
    Af = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = rng.standard_normal(d)
    y = Af @ truth + 0.3 * rng.standard_normal(m)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    cov = 0.09 * np.eye(m)
    posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), tda.LinearModel(Af + 0.1 * rng.standard_normal((m, d)) / np.sqrt(d))),
             tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), tda.LinearModel(Af + 0.05 * rng.standard_normal((m, d)) / np.sqrt(d))),
             tda.Posterior(prior, tda.GaussianLogLike(y, cov), tda.LinearModel(Af))]
    th0 = [truth + 0.1 * rng.standard_normal(d) for _ in range(N)]
    res = tda.sample(posts, tda.GaussianRandomWalk(0.004 * np.eye(d)), 25, n_chains=N, initial_parameters=th0, subchain_length=[3, 2],
                     adaptive_error_model="state-independent", seed=11)
    assert res["sampler"] == "MLDA" and res.get("backend", "hip") != "host"
    fine = res["chain_l2_3"]
    assert len(fine) == 26
    link = fine[-1]
    ref = posts[2].create_link(link.parameters)
    assert np.isclose(link.posterior, ref.posterior, rtol=1e-10)
    acc = np.mean([np.mean(res["chain_l2_%d" % i].accepted[1:]) for i in range(N)])
    assert 0.05 < acc <= 1.0
    da = tda.sample(posts[1:], tda.CrankNicolson(scaling=0.05), 20, n_chains=N, initial_parameters=th0, subchain_length=1,
                    adaptive_error_model="state-dependent", seed=12)
    assert da["sampler"] == "DA" and da.get("backend", "hip") != "host"
    lk = da["chain_fine_0"][-1]
    assert np.isclose(lk.posterior, posts[2].create_link(lk.parameters).posterior, rtol=1e-10)
