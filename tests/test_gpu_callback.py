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


def _level_model(k):
    """Three fidelities of the same non-linear model: coarser levels drop part of the interaction term and perturb W."""
    def fn(theta):
        theta = np.atleast_2d(theta)
        W = 0.1 + 0.01 * ((np.arange(M)[:, None] * 7 + np.arange(theta.shape[1])[None, :] * 3) % 11) + 0.004 * (2 - k)
        return np.tanh(theta @ W.T) + 0.25 * (0.6 + 0.2 * k) * theta[:, [0]] * theta[:, [-1]]
    return fn


@pytest.mark.parametrize("case", ["da_pcn", "da_grw_diag", "mlda_am"])
def test_callback_hierarchy_matches_oracle(case):
    """Delayed Acceptance / MLDA with every level behind a batched host callback (host-sequenced level actions,
    k_ext_level_action) against the oracle's DAChain / MLDAChain restatement running the same NumPy models; engine on its
    own Philox stream, base-level normals exported, uniforms regenerated by the oracle's Philox."""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, N = 6, 19
    rng = np.random.default_rng(15)
    truth = 0.5 * rng.standard_normal(d)
    if case == "mlda_am":
        nl, sl, n_fine, block = 3, [3, 2], 14, 7
    else:
        nl, sl, n_fine, block = 2, [4], 25, 0
    models = [_level_model(k + (3 - nl)) for k in range(nl)]
    y = models[-1](truth)[0] + 0.05 * rng.standard_normal(M)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    diag = case == "da_grw_diag"
    noise = 0.05 ** 2 * (1.0 + 0.1 * np.arange(M)) if diag else 0.05 ** 2
    seed = 4711
    calls = [[] for _ in range(nl)]
    e = Engine(N, d, seed=seed, n_levels=nl, block_steps=block)
    e.set_prior(pm, np.diag(pv))
    for k in range(nl):
        def fn(thetas, k=k):
            calls[k].append(thetas.shape)
            return models[k](thetas)
        e.set_level_callback(k, fn, y, 1 if diag else 0, noise if diag else [noise])
    if case == "da_pcn":
        e.set_proposal(1, None, scaling=0.04)
        prop = dict(kind="pcn", scaling=0.04)
    elif case == "da_grw_diag":  # adaptive scaling: the window of accept flags contains the alignment entries of the fine level
        e.set_proposal(0, 2e-3 * np.eye(d), scaling=1.0, adaptive=True, gamma=1.02, period=15)
        prop = dict(kind="grw", C=2e-3 * np.eye(d), scaling=1.0, adaptive=True, gamma=1.02, period=15)
    else:
        e.set_proposal(2, 2e-3 * np.eye(d), t0=20, period=10, adaptive=True, gamma=1.02)
        prop = dict(kind="am", C0=2e-3 * np.eye(d), t0=20, period=10, adaptive=True, gamma=1.02)
    e.set_subchains(sl, False)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    scal = e.proposal_state()["scaling"]
    e.close()
    # one call per level for the initial links (level 0 twice: the single-level and the hierarchy initialisation), then one per local step
    for k in range(nl):
        assert all(c == (N, d) for c in calls[k]) and len(calls[k]) >= rows[k] + 1
    us, _ = _oracle_uniforms(seed, N, rows, sl)
    prior = orc.MVNPrior(pm, np.diag(pv))
    levels = [orc.CallableGaussianLevel(models[k], y, "diag" if diag else "iso", noise, prior) for k in range(nl)]
    res, pstate = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, None)
    for k in range(nl):
        ref = res[k]
        sk = slice(1, None) if k == nl - 1 else slice(None)
        assert np.array_equal(outs[k][2], ref["accepted"][:, sk].T), "level %d accept masks differ" % k
        np.testing.assert_allclose(outs[k][1][:, :, 2], ref["logpost"][:, sk].T, rtol=1e-10)
        np.testing.assert_allclose(outs[k][0], np.swapaxes(ref["theta"][:, sk], 0, 1), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(scal, pstate.scaling, rtol=1e-12)
    assert 0.05 < outs[nl - 1][2].mean() < 0.98


def test_callback_hierarchy_through_sample_api():
    import tinyda_amd as tda

    d = 6
    rng = np.random.default_rng(3)
    truth = 0.3 * rng.standard_normal(d)
    y = _level_model(2)(truth)[0] + 0.05 * rng.standard_normal(M)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    like = tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(M))
    posts = [tda.Posterior(prior, like, tda.BatchedModel(_level_model(k), M)) for k in (1, 2)]
    th0 = [truth + 0.05 * rng.standard_normal(d) for _ in range(8)]
    res = tda.sample(posts, tda.CrankNicolson(scaling=0.04), 30, n_chains=8, initial_parameters=th0, subchain_length=3, seed=9)
    assert res["sampler"] == "DA" and res.get("backend", "hip") != "host"
    assert len(res["chain_fine_2"]) == 31 and len(res["chain_coarse_2"]) == 90
    link = res["chain_fine_5"][-1]
    assert np.isclose(link.posterior, posts[1].create_link(link.parameters).posterior, rtol=1e-10)
    ada = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, 0.05 ** 2 * np.eye(M)), tda.BatchedModel(_level_model(1), M)), posts[1]]
    aem = tda.sample(ada, tda.CrankNicolson(scaling=0.04), 12, n_chains=4, subchain_length=2, adaptive_error_model="state-independent",
                     initial_parameters=th0[:4], seed=2, backend="hip")  # the state-independent error model runs behind callbacks too
    lk = aem["chain_fine_1"][-1]
    assert np.isclose(lk.posterior, posts[1].create_link(lk.parameters).posterior, rtol=1e-10)
    dep = tda.sample(ada, tda.CrankNicolson(scaling=0.04), 12, n_chains=4, subchain_length=1, adaptive_error_model="state-dependent",
                     initial_parameters=th0[:4], seed=3, backend="hip")
    lk = dep["chain_fine_2"][-1]
    assert np.isclose(lk.posterior, posts[1].create_link(lk.parameters).posterior, rtol=1e-10)
    rnd = tda.sample(posts, tda.CrankNicolson(scaling=0.04), 12, n_chains=4, subchain_length=3, randomize_subchain_length=True,
                     initial_parameters=th0[:4], seed=4, backend="hip")
    lk = rnd["chain_fine_0"][-1]
    assert np.isclose(lk.posterior, posts[1].create_link(lk.parameters).posterior, rtol=1e-10)
    with pytest.raises(tda.EngineError):  # ... but not together with an error model (nor is it for linear levels): the engine says so
        tda.sample(ada, tda.CrankNicolson(scaling=0.04), 5, n_chains=4, subchain_length=3, randomize_subchain_length=True,
                   adaptive_error_model="state-independent", backend="hip")


def test_multilevel_sampling_with_plain_python_models():
    """The reference's everyday call -- sample([coarse, fine], ...) with plain Python callables as models -- runs on the
    device engine: the callables are evaluated chain by chain behind the batched-callback interface."""
    import tinyda_amd as tda

    d = 6
    rng = np.random.default_rng(7)
    truth = 0.3 * rng.standard_normal(d)
    fine = lambda th: _level_model(2)(th)[0]  # noqa: E731  theta -> ndarray, one chain at a time
    coarse = lambda th: _level_model(1)(th)[0]  # noqa: E731
    y = fine(truth) + 0.05 * rng.standard_normal(M)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    cov = 0.05 ** 2 * np.eye(M)
    posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), coarse), tda.Posterior(prior, tda.GaussianLogLike(y, cov), fine)]
    th0 = [truth + 0.05 * rng.standard_normal(d) for _ in range(6)]
    res = tda.sample(posts, tda.GaussianRandomWalk(2e-3 * np.eye(d), adaptive=True, period=20), 40, n_chains=6, initial_parameters=th0,
                     subchain_length=3, adaptive_error_model="state-independent", seed=8)
    assert res["sampler"] == "DA" and res.get("backend", "hip") != "host"
    link = res["chain_fine_4"][-1]
    assert np.isclose(link.posterior, posts[1].create_link(link.parameters).posterior, rtol=1e-10)
    assert np.allclose(link.model_output, fine(link.parameters))
    assert np.mean([np.mean(res["chain_fine_%d" % i].accepted[1:]) for i in range(6)]) > 0.05


@pytest.mark.parametrize("model_kind", ["callback", "source"])
def test_dense_noise_over_external_models(model_kind):
    """DefaultGaussianLogLike (correlated data noise) over a callback and over a source-defined model: single-level AM against
    the oracle, and as the fine level of a Delayed-Acceptance pair."""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tests.test_gpu_usermodel import SRC, np_model as src_model
    from tinyda_amd.engine import Engine

    if model_kind == "callback":
        d, m, model = 6, M, np_model
    else:
        d, m, model = 5, 23, src_model
    N, T = 19, 120
    rng = np.random.default_rng(33)
    truth = 0.5 * rng.standard_normal(d)
    y = model(truth)[0] + 0.05 * rng.standard_normal(m)
    Ln = 0.05 * np.eye(m) + 0.01 * np.tril(rng.standard_normal((m, m)))
    cov = Ln @ Ln.T
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    e = Engine(N, d, seed=17, block_steps=32)
    e.set_prior(pm, np.diag(pv))
    if model_kind == "callback":
        e.set_level_callback(0, model, y, 2, cov)
    else:
        e.set_level_source(0, SRC, y, 2, cov)
    e.set_proposal(2, 2e-3 * np.eye(d), t0=40, period=20)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    e.close()
    prior = orc.MVNPrior(pm, np.diag(pv))
    lvl = orc.CallableGaussianLevel(model, y, "dense", cov, prior)
    ref = orc.run_mh(lvl, dict(kind="am", C0=2e-3 * np.eye(d), t0=40, period=20), theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    assert 0.05 < acc.mean() < 0.95
    # Delayed Acceptance: isotropic coarse level, dense fine level
    seed, sl, n_fine = 29, [3], 20
    e = Engine(N, d, seed=seed, n_levels=2)
    e.set_prior(pm, np.diag(pv))
    for k in range(2):
        kind, nz = (0, [0.08 ** 2]) if k == 0 else (2, cov)
        if model_kind == "callback":
            e.set_level_callback(k, model, y, kind, nz)
        else:
            e.set_level_source(k, SRC, y, kind, nz)
    e.set_proposal(1, None, scaling=0.04)
    e.set_subchains(sl, False)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    e.close()
    us, _ = _oracle_uniforms(seed, N, rows, sl)
    levels = [orc.CallableGaussianLevel(model, y, "iso", 0.08 ** 2, prior), orc.CallableGaussianLevel(model, y, "dense", cov, prior)]
    res, _ = orc.run_multilevel(levels, dict(kind="pcn", scaling=0.04), sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, None)
    for i in range(2):
        sk = slice(1, None) if i == 1 else slice(None)
        assert np.array_equal(outs[i][2], res[i]["accepted"][:, sk].T), "level %d accept masks differ" % i
        np.testing.assert_allclose(outs[i][1][:, :, 2], res[i]["logpost"][:, sk].T, rtol=1e-10)


@pytest.mark.parametrize("kind,model_kind", [("indep", "callback"), ("indep", "source"), ("owcn", "callback"), ("owcn", "source")])
def test_independence_and_operator_weighted_pcn_over_external_models(kind, model_kind):
    """IndependenceSampler and OperatorWeightedCrankNicolson proposals over a callback / source-defined model against the oracle."""
    from tests.test_gpu_usermodel import SRC, np_model as src_model
    from tinyda_amd.engine import Engine

    if model_kind == "callback":
        d, m, model = 6, M, np_model
    else:
        d, m, model = 5, 23, src_model
    N, T = 21, 130
    rng = np.random.default_rng(44)
    truth = 0.5 * rng.standard_normal(d)
    y = model(truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.03 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    e = Engine(N, d, seed=23, block_steps=40)
    e.set_prior(pm, np.diag(pv))
    if model_kind == "callback":
        e.set_level_callback(0, model, y, 0, [0.05 ** 2])
    else:
        e.set_level_source(0, SRC, y, 0, [0.05 ** 2])
    if kind == "indep":
        q_mean, q_cov = truth + 0.01 * rng.standard_normal(d), 4e-4 * np.eye(d)
        e.set_proposal(4, q_cov, q_mean=q_mean)
        prop = dict(kind="indep", q_mean=q_mean, q_cov=q_cov)
    else:
        Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        B = Q @ np.diag(np.linspace(0.05, 0.6, d)) @ Q.T
        S, Nop = orc.owcn_operators(B, 0.004)
        e.set_proposal(5, None, state_operator=S, noise_operator=Nop)
        prop = dict(kind="owcn", B=B, scaling=0.004)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    e.close()
    lvl = orc.CallableGaussianLevel(model, y, "iso", 0.05 ** 2, orc.MVNPrior(pm, np.diag(pv)))
    ref = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    assert 0.02 < acc.mean() < 0.98
