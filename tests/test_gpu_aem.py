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


@pytest.mark.parametrize("name", ["g8_da_aem_indep", "g8_da_aem_dep", "g8_da_aem_dep_pcn", "g8_da_aem_indep_m72", "g8_da_aem_dep_pcn_m128",
                                  "g8_da_aem_indep_m200", "g8_da_aem_dep_pcn_m256",
                                  "g8_da_aem_indep_d80", "g8_da_aem_dep_pcn_d96"])
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


@pytest.mark.parametrize("name", ["g8_mlda_aem", "g8_mlda_aem_m100", "g8_mlda_aem_m160", "g8_mlda_aem_d72"])
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


@pytest.mark.parametrize("m", [100, 200])
def test_sample_api_with_error_model_beyond_64_outputs(m):
    """tda.sample(..., adaptive_error_model=...) lowers AdaptiveGaussianLogLike levels with up to 256 outputs (two- and four-wave
    error-model kernels; beyond 128: k_aem_refresh_big); the finest-level links carry the posterior of the finest model."""
    import scipy.stats as st

    import tinyda_amd as tda

    d, N = 6, 12
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


def _diag_levels(rng, d, m, nl, sigma, diag_noise):
    from oracle import tinyda_oracle as orc

    truth = rng.standard_normal(d) * 0.5
    Af = rng.standard_normal((m, d)) / np.sqrt(d)
    y = Af @ truth + sigma * rng.standard_normal(m)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    var = sigma ** 2 * (1.0 + (rng.random(m) if diag_noise else np.zeros(m)))
    spec, As, bs = [], [], []
    for k in range(nl):
        A = Af + 0.08 * (nl - 1 - k) * rng.standard_normal((m, d)) / np.sqrt(d)
        b = 0.05 * (nl - 1 - k) * rng.standard_normal(m)
        As.append(A)
        bs.append(b)
        if k < nl - 1:
            spec.append(dict(A=A, b=b, y=y, prior=prior, cov=np.diag(var)))
        else:
            spec.append(dict(A=A, b=b, y=y, prior=prior, var=float(var[0])))
    return spec, As, bs, y, var, truth


@pytest.mark.parametrize("nl,sl,d,m,diag_noise,kind", [(2, [3], 5, 8, False, "grw"), (2, [4], 6, 70, True, "pcn"), (3, [3, 2], 5, 24, False, "grw"),
                                                        (3, [2, 2], 8, 300, False, "am"),
                                                        # more than 32 parameters and at most 128 outputs: the base subchains run in
                                                        # k_da_steps with per-chain corrected data and inverse variances
                                                        # (AdaptiveMetropolis with t0 = 4 in 40 dimensions factors a rank-4 + 1e-6 I covariance:
                                                        # device and LAPACK Cholesky then differ by 1e-10 in the proposals, so these cases
                                                        # use the other proposals; AM on this path: tests/test_gpu_switches.py)
                                                        (3, [5, 3], 40, 128, False, "grw"), (2, [4], 64, 100, True, "pcn"), (3, [3, 2], 33, 50, True, "grw")])
def test_diagonal_error_model_vs_oracle(eng_mod, nl, sl, d, m, diag_noise, kind):
    """TDA_AEM_STATE_INDEPENDENT_DIAGONAL (extension: diagonal bias covariances, any output dimension -- here up to m = 300,
    beyond the 128 of the dense model): Delayed Acceptance and MLDA on the engine's own Philox stream against the oracle's
    restatement (run_multilevel_aem(..., diagonal=True)): accept flags of every level equal, finest log-posterior to 1e-10,
    stacked biases and inverse variances of the coarse levels to 1e-9."""
    from oracle import tinyda_oracle as orc

    rng = np.random.default_rng(100 * nl + m)
    N, n_fine = 12, 14
    sigma = 0.3
    spec, As, bs, y, var, truth = _diag_levels(rng, d, m, nl, sigma, diag_noise)
    if diag_noise:  # the finest level of the oracle's error-model chains is isotropic
        spec[-1]["var"] = float(var[0])
    theta0 = truth + 0.1 * rng.standard_normal((N, d))
    e = eng_mod.Engine(N, d, seed=21, n_levels=nl)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(nl):
        if k < nl - 1 and diag_noise:
            e.set_level(k, As[k], y, 1, var, b=bs[k])
        else:
            e.set_level(k, As[k], y, 0, float(var[0]), b=bs[k])
    if kind == "grw":
        prop = dict(kind="grw", C=0.01 * np.eye(d), scaling=1.0, adaptive=True, gamma=1.02, period=5)
        e.set_proposal(0, prop["C"], scaling=1.0, adaptive=True, gamma=1.02, period=5)
    elif kind == "pcn":
        prop = dict(kind="pcn", scaling=0.15)
        e.set_proposal(1, None, scaling=0.15)
    else:
        prop = dict(kind="am", C0=0.01 * np.eye(d), t0=4, period=4, sd=min(1.0, 2.4 ** 2 / d), epsilon=1e-6)
        e.set_proposal(2, prop["C0"], t0=4, period=4)
    e.set_subchains(sl)
    e.set_error_model("state-independent-diagonal")
    e.init(theta0)
    T0 = n_fine * int(np.prod(sl))
    z, u0 = e.set_export(T0)
    outs = e.run_levels_host(n_fine)
    # the oracle on the engine's variates; upper-level uniforms from the RNG contract (stream 1, block = level)
    st = orc.PhiloxStream(21)
    rows = e.rows_per_level(n_fine)
    us = [np.swapaxes(u0, 0, 1)] + [np.stack([st.uniform(np.arange(N), t, level=k) for t in range(rows[k])], axis=1) for k in range(1, nl)]
    ref, state = orc.run_multilevel_aem(spec, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, "state-independent", diagonal=True)
    for k in range(nl):
        p, s_, a = outs[k]
        off = 1 if k == nl - 1 else 0
        assert np.array_equal(a, ref[k]["accepted"][:, off:].T), "level %d: %d accept flips" % (k, int((a != ref[k]["accepted"][:, off:].T).sum()))
        np.testing.assert_allclose(p, np.swapaxes(ref[k]["theta"][:, off:], 0, 1), rtol=1e-10, atol=1e-12)
        if k == nl - 1:
            np.testing.assert_allclose(s_[:, :, 2], ref[k]["logpost"][:, off:].T, rtol=RTOL)
    for k in range(nl - 1):
        bias, P = e.error_model_state(k, m)
        np.testing.assert_allclose(bias, state["bias"][k], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.einsum("nii->ni", P), np.einsum("nii->ni", state["cov_inv"][k]), rtol=1e-9)
    e.close()


def test_sample_api_with_diagonal_error_model():
    """tda.sample(..., adaptive_error_model='state-independent', error_model_covariance='diagonal') with 512 outputs per level"""
    import scipy.stats as st

    import tinyda_amd as tda

    d, m, N = 6, 512, 16
    rng = np.random.default_rng(6)
    Af = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = rng.standard_normal(d)
    y = Af @ truth + 0.3 * rng.standard_normal(m)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    cov = 0.09 * np.eye(m)
    posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), tda.LinearModel(Af + 0.1 * rng.standard_normal((m, d)) / np.sqrt(d))),
             tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), tda.LinearModel(Af + 0.05 * rng.standard_normal((m, d)) / np.sqrt(d))),
             tda.Posterior(prior, tda.GaussianLogLike(y, cov), tda.LinearModel(Af))]
    th0 = [truth + 0.05 * rng.standard_normal(d) for _ in range(N)]
    res = tda.sample(posts, tda.GaussianRandomWalk(0.001 * np.eye(d)), 20, n_chains=N, initial_parameters=th0, subchain_length=[3, 2],
                     adaptive_error_model="state-independent", error_model_covariance="diagonal", seed=4, backend="hip")
    assert res["sampler"] == "MLDA" and res["backend"] == "hip" and len(res["chain_l2_0"]) == 21
    link = res["chain_l2_5"][-1]
    assert np.isclose(link.posterior, posts[2].create_link(link.parameters).posterior, rtol=1e-10)
    with pytest.raises(tda.EngineError):  # the dense model stops at 128 outputs on the device
        tda.sample(posts, tda.GaussianRandomWalk(0.001 * np.eye(d)), 5, n_chains=N, initial_parameters=th0, subchain_length=[3, 2],
                   adaptive_error_model="state-independent", seed=4, backend="hip")
