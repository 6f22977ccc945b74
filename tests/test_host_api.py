"""Host mirror of the reference API: classes, factory dispatch, error behaviour, result layout, and the
host chain driver replaying the reference's golden variates."""
import numpy as np
import pytest
import scipy.stats as stats

import tinyda_amd as tda


def test_factory_dispatch_and_errors(golden):
    g = golden("g3_loglike_kats")
    data, X = g["data"], g["X"]
    m = len(data)
    iso = tda.GaussianLogLike(data, float(g["iso_var"]) * np.eye(m))
    diag = tda.GaussianLogLike(data, g["diag_cov"])
    dense = tda.GaussianLogLike(data, g["dense_cov"])
    assert [type(o).__name__ for o in (iso, diag, dense)] == list(g["class_names"])
    for obj, key in ((iso, "iso"), (diag, "diag"), (dense, "dense")):
        np.testing.assert_allclose([obj.loglike(x) for x in X], g["out_" + key], rtol=1e-12)
        np.testing.assert_allclose([obj.grad_loglike(x) for x in X], g["grad_" + key], rtol=1e-11, atol=1e-13)
    with pytest.raises(TypeError):
        tda.GaussianLogLike(data, [[1.0]])
    with pytest.raises(TypeError):
        tda.GaussianLogLike(data, np.ones(m))
    with pytest.raises(ValueError):
        tda.GaussianLogLike(data, np.eye(m + 1))
    with pytest.raises(ValueError):
        tda.GaussianLogLike(data, np.ones((m, m + 1)))
    ada = tda.AdaptiveGaussianLogLike(data, g["dense_cov"])
    np.testing.assert_allclose([ada.loglike(x) for x in X], g["out_ada0"], rtol=1e-12)
    ada.set_bias(g["bias"], g["bias_cov"])
    np.testing.assert_allclose([ada.loglike(x) for x in X], g["out_ada1"], rtol=1e-12)
    np.testing.assert_allclose([ada.loglike_custom_bias(x, g["custom_bias"]) for x in X], g["out_ada_custom"], rtol=1e-12)
    ada2 = tda.AdaptiveGaussianLogLike(data, g["dense_cov"])
    ada2.set_bias(g["bias"], g["tiny_cov"])
    np.testing.assert_allclose([ada2.loglike(x) for x in X], g["out_ada_tiny"], rtol=1e-12)
    ada2.set_bias(g["bias"], g["mixed_cov"])
    np.testing.assert_allclose([ada2.loglike(x) for x in X], g["out_ada_mixed"], rtol=1e-12)


def test_recursive_moments_bit_identical(golden):
    g = golden("g7_moments")
    X = g["X"]
    r = tda.RecursiveSampleMoments(X[0].copy(), np.zeros((X.shape[1],) * 2), sd=float(g["sd"]), epsilon=float(g["epsilon"]))
    for i, x in enumerate(X[1:]):
        r.update(x)
        assert np.array_equal(r.get_mu(), g["mu_hist"][i]) and np.array_equal(r.get_sigma(), g["sigma_hist"][i])
    z = tda.ZeroMeanRecursiveSampleMoments(np.zeros((X.shape[1],) * 2))
    for i, x in enumerate(X):
        z.update(x)
        assert np.array_equal(z.get_sigma(), g["zero_mean_hist"][i])


def test_posterior_and_link_protocol():
    A = np.array([[1.0, 2.0], [0.5, -1.0], [0.0, 3.0]])
    prior = stats.multivariate_normal(np.zeros(2), np.eye(2))
    post = tda.Posterior(prior, tda.GaussianLogLike(np.ones(3), 0.5 * np.eye(3)), lambda th: (A @ th, th.sum()))
    link = post.create_link(np.array([0.2, -0.1]))
    assert link.qoi == pytest.approx(0.1) and link.posterior == link.prior + link.likelihood
    assert post(np.array([0.2, -0.1])) == link.posterior == post.logpdf(np.array([0.2, -0.1]))
    with pytest.raises(TypeError):
        tda.Posterior(prior, tda.GaussianLogLike(np.ones(3), np.eye(3)), lambda th: [1, 2, 3]).create_link(np.zeros(2))
    with pytest.raises(TypeError):
        tda.GaussianRandomWalk([[1.0]])
    with pytest.raises(ValueError):
        tda.AdaptiveMetropolis(np.ones((2, 3)))
    assert tda.AdaptiveMetropolis(np.eye(64)).sd == pytest.approx(2.4 ** 2 / 64)
    with pytest.raises(TypeError):  # pCN needs a scipy multivariate normal prior (sampler.py:138-143)
        tda.sample(tda.Posterior(stats.norm(), tda.GaussianLogLike(np.ones(3), np.eye(3)), lambda th: A @ th),
                   tda.CrankNicolson(), 5, backend="host")


class _Replay:
    """Feeds recorded variates to the host proposals / chain (np.random.standard_normal, np.random.random)."""

    def __init__(self, z, u):
        self.z, self.u, self.iz, self.iu = z, u, 0, 0

    def __enter__(self):
        self.saved = (np.random.standard_normal, np.random.random)
        np.random.standard_normal = self._normal
        np.random.random = self._uniform
        return self

    def __exit__(self, *a):
        np.random.standard_normal, np.random.random = self.saved

    def _normal(self, n):
        self.iz += 1
        return self.z[self.iz - 1]

    def _uniform(self):
        self.iu += 1
        return self.u[self.iu - 1]


@pytest.mark.parametrize("name", ["g1_basic_sampler", "g2_am_small_adaptive", "g2b_pcn"])
def test_host_chain_replays_reference(golden, name):
    g = golden(name)
    prior = stats.multivariate_normal(g["prior_mean"], g["prior_cov"])
    if "noise_var" in g.files:
        cov = float(g["noise_var"]) * np.eye(len(g["data"]))
    else:
        cov = np.diag(g["noise_cov"])
    A = g["A"]
    post = tda.Posterior(prior, tda.GaussianLogLike(g["data"], cov), lambda th: A @ th)
    for c in range(g["theta"].shape[0]):
        if name.startswith("g1"):
            prop = tda.GaussianRandomWalk(g["C"], scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]),
                                          period=int(g["period"]))
        elif name.startswith("g2b"):
            prop = tda.CrankNicolson(scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]),
                                     period=int(g["period"]))
        else:
            prop = tda.AdaptiveMetropolis(g["C0"], sd=float(g["sd"]), epsilon=float(g["epsilon"]), t0=int(g["t0"]),
                                          period=int(g["period"]), adaptive=bool(g["adaptive"]), gamma=float(g["gamma"]))
        with _Replay(g["z"][c], g["u"][c]):
            ch = tda.Chain(post, prop, g["theta0"][c].copy())
            ch.sample(g["z"].shape[1], progressbar=False)
        assert np.array_equal(np.array(ch.accepted, dtype=np.uint8), g["accepted"][c])
        np.testing.assert_allclose([l.posterior for l in ch.chain], g["logpost"][c], rtol=1e-10)


def test_config1_basic_sampler_host_path():
    """BASELINE config 1: opaque Python model, 2 chains x 2000 iterations, result layout of sampler.py:305-309."""
    rs = np.random.RandomState(0)
    x = np.linspace(0, 1, 50)
    y = 1 + 2 * x + rs.normal(0, 0.2, 50)
    model = lambda th: th[0] + th[1] * x  # noqa: E731  (opaque closure -> host protocol)
    post = tda.Posterior(stats.multivariate_normal(np.zeros(2), np.eye(2)), tda.GaussianLogLike(y, 0.04 * np.eye(50)), model)
    np.random.seed(3)
    res = tda.sample(post, tda.GaussianRandomWalk(np.eye(2), scaling=0.1, adaptive=True), 2000, n_chains=2,
                     force_sequential=True)
    assert res["sampler"] == "MH" and res["n_chains"] == 2 and res["iterations"] == 2001 and res["backend"] == "host"
    assert len(res["chain_0"]) == 2001 and isinstance(res["chain_1"][5], tda.Link)
    s = tda.get_samples(res, burnin=500)
    assert s["iterations"] == 1501 and s["dimension"] == 2 and s["chain_1"].shape == (1501, 2)
    st = tda.get_samples(res, "stats")
    np.testing.assert_allclose(st["chain_0"][:, 2], st["chain_0"][:, 0] + st["chain_0"][:, 1])
    mo = tda.get_samples(res, "model_output")
    assert mo["dimension"] == 50
    pooled = np.concatenate([s["chain_0"], s["chain_1"]])
    # conjugate posterior mean of the linear-Gaussian problem
    Ad = np.stack([np.ones_like(x), x], axis=1)
    cov = np.linalg.inv(Ad.T @ Ad / 0.04 + np.eye(2))
    mean = cov @ (Ad.T @ y / 0.04)
    assert np.all(np.abs(pooled.mean(axis=0) - mean) < 5 * np.sqrt(np.diag(cov)) / np.sqrt(30)), (pooled.mean(axis=0), mean)
    # to_inference_data (diagnostics.py:6-70): groups, variable names and (chain, draw) layout; ArviZ objects when installed,
    # the light container otherwise
    idata = tda.to_inference_data(res, burnin=500, parameter_names=["intercept", "slope"])
    assert list(idata.posterior) == ["intercept", "slope"] or set(idata.posterior.data_vars) == {"intercept", "slope"}
    assert np.asarray(idata.posterior["slope"]).shape == (2, 1501)
    np.testing.assert_array_equal(np.asarray(idata.posterior["slope"])[1], s["chain_1"][:, 1])
    assert np.asarray(idata.posterior_predictive["obs_49"]).shape == (2, 1501)
    np.testing.assert_allclose(np.asarray(idata.sample_stats["posterior"]),
                               np.asarray(idata.sample_stats["prior"]) + np.asarray(idata.sample_stats["likelihood"]))
    assert len(idata.qoi) == 0  # the model returns no quantity of interest
    if hasattr(idata, "summary"):
        row = idata.summary()["slope"]
        assert abs(row["mean"] - mean[1]) < 0.2 and row["ess_bulk"] > 20 and 0.9 < row["r_hat"] < 1.5
        # the columns of az.summary (examples/Basic Sampler.ipynb cell 17)
        assert set(row) == {"mean", "sd", "hdi_3%", "hdi_97%", "mcse_mean", "ess_bulk", "ess_tail", "r_hat"}
        assert row["hdi_3%"] < row["mean"] < row["hdi_97%"] and row["ess_tail"] > 10 and 0 < row["mcse_mean"] < row["sd"]


def test_lowering_pass():
    from tinyda_amd.api import _device_plan

    A = np.random.default_rng(0).standard_normal((10, 4))
    prior = stats.multivariate_normal(np.zeros(4), np.eye(4))
    like = tda.GaussianLogLike(np.zeros(10), 0.1 * np.eye(10))
    assert _device_plan([tda.Posterior(prior, like, tda.LinearModel(A))], tda.AdaptiveMetropolis(np.eye(4))) is not None
    assert _device_plan([tda.Posterior(prior, like, lambda th: A @ th)], tda.AdaptiveMetropolis(np.eye(4))) is None
    assert _device_plan([tda.Posterior(stats.norm(), like, tda.LinearModel(A))], tda.GaussianRandomWalk(np.eye(4))) is None

    class Custom(tda.GaussianRandomWalk):
        pass

    assert _device_plan([tda.Posterior(prior, like, tda.LinearModel(A))], Custom(np.eye(4))) is None


def test_host_fallback_is_announced_with_the_rule_that_refused_the_problem():
    """backend='auto' past a device limit: the reference's protocol runs, and ONE HostFallbackWarning names why;
    backend='host' is silent, backend='hip' raises with the same reason"""
    import warnings

    from tinyda_amd import api

    d = 129  # one parameter past the engine's limit (0.5: 128 for single-level chains)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((8, d))
    post = tda.Posterior(stats.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(np.zeros(8), 0.1 * np.eye(8)), tda.LinearModel(A))
    with pytest.warns(tda.HostFallbackWarning, match="more than 128 parameters") as rec:
        res = tda.sample(post, tda.GaussianRandomWalk(np.eye(d), scaling=0.05), 5, n_chains=1)
    assert res["backend"] == "host" and len([w for w in rec if issubclass(w.category, tda.HostFallbackWarning)]) == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        tda.sample(post, tda.GaussianRandomWalk(np.eye(d), scaling=0.05), 5, n_chains=1, backend="host")
    with pytest.raises(tda.EngineError, match="more than 128 parameters"):
        tda.sample(post, tda.GaussianRandomWalk(np.eye(d), scaling=0.05), 5, n_chains=1, backend="hip")
    # every refusal of the lowering pass leaves its reason
    prior4 = stats.multivariate_normal(np.zeros(4), np.eye(4))
    like = tda.GaussianLogLike(np.zeros(10), 0.1 * np.eye(10))
    assert api._device_plan([tda.Posterior(prior4, like, lambda th: th)], tda.AdaptiveMetropolis(np.eye(4))) is None
    assert "opaque Python model" in api._refusal[0]
    mo = api.MAX_AEM_OUTPUTS + 1  # (0.5: 256 outputs for the dense error model over linear levels)
    posts = [tda.Posterior(prior4, tda.AdaptiveGaussianLogLike(np.zeros(mo), 0.1 * np.eye(mo)), tda.LinearModel(rng.standard_normal((mo, 4)))),
             tda.Posterior(prior4, tda.GaussianLogLike(np.zeros(mo), 0.1 * np.eye(mo)), tda.LinearModel(rng.standard_normal((mo, 4))))]
    assert api._device_plan(posts, tda.CrankNicolson(), error_model="state-independent") is None and "AdaptiveGaussianLogLike" in api._refusal[0]
    mo = 200  # ... and inside the limit also below DREAM(Z) (host-sequenced level actions: k_ext_aem_*<256>)
    posts = [tda.Posterior(prior4, tda.AdaptiveGaussianLogLike(np.zeros(mo), 0.1 * np.eye(mo)), tda.LinearModel(rng.standard_normal((mo, 4)))),
             tda.Posterior(prior4, tda.GaussianLogLike(np.zeros(mo), 0.1 * np.eye(mo)), tda.LinearModel(rng.standard_normal((mo, 4))))]
    assert api._device_plan(posts, tda.CrankNicolson(), error_model="state-independent") is not None
    assert api._device_plan(posts, tda.DREAMZ(M0=20), error_model="state-independent") is not None


def test_dense_error_model_with_a_dense_fine_level_falls_back_with_the_warning():
    """AdaptiveGaussianLogLike coarse level + dense observation covariance on the finest level under the dense error
    model is a legal reference configuration the engine does not lower (its finest level must be isotropic): the lowering pass
    refuses it, so backend='auto' runs the host protocol with ONE HostFallbackWarning instead of raising from tda_engine_init."""
    from tinyda_amd import api

    d, m = 3, 6
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, d))
    y = A @ np.ones(d)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    L = 0.3 * np.eye(m) + 0.05 * np.tril(rng.standard_normal((m, m)))
    posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, 0.1 * np.eye(m)), tda.LinearModel(A + 0.05)),
             tda.Posterior(prior, tda.GaussianLogLike(y, L @ L.T), tda.LinearModel(A))]
    assert api._device_plan(posts, tda.CrankNicolson(scaling=0.2)) is not None  # without an error model: lowered (0.4)
    for kind in ("state-independent", "state-dependent"):
        assert api._device_plan(posts, tda.CrankNicolson(scaling=0.2), False, kind) is None
        assert "finest level must have an isotropic likelihood" in api._refusal[0]
    with pytest.warns(tda.HostFallbackWarning, match="finest level must have an isotropic likelihood") as rec:
        res = tda.sample(posts, tda.CrankNicolson(scaling=0.2), 6, n_chains=1, subchain_length=2, adaptive_error_model="state-independent")
    assert res["backend"] == "host" and res["sampler"] == "DA"
    assert len([w for w in rec if issubclass(w.category, tda.HostFallbackWarning)]) == 1
    with pytest.raises(tda.EngineError, match="finest level must have an isotropic likelihood"):
        tda.sample(posts, tda.CrankNicolson(scaling=0.2), 6, n_chains=1, subchain_length=2, adaptive_error_model="state-independent", backend="hip")
    iso = [posts[0], tda.Posterior(prior, tda.GaussianLogLike(y, 0.1 * np.eye(m)), tda.LinearModel(A))]
    assert api._device_plan(iso, tda.CrankNicolson(scaling=0.2), False, "state-independent") is not None


def test_multilevel_lowering_and_argument_checks():
    from tinyda_amd.api import _device_plan

    rng = np.random.default_rng(1)
    prior = stats.multivariate_normal(np.zeros(4), np.eye(4))
    posts = [tda.Posterior(prior, tda.GaussianLogLike(np.zeros(m), 0.1 * np.eye(m)), tda.LinearModel(rng.standard_normal((m, 4))))
             for m in (6, 12, 20)]
    assert _device_plan(posts, tda.CrankNicolson()) is not None
    other = tda.Posterior(stats.multivariate_normal(np.ones(4), np.eye(4)), posts[1].likelihood, posts[1].model)
    assert _device_plan([posts[0], other], tda.CrankNicolson()) is None  # priors must agree across levels
    with pytest.raises(ValueError):
        tda.sample(posts[:2], tda.CrankNicolson(), 5, subchain_length=1, randomize_subchain_length=True)
    with pytest.raises(ValueError):
        tda.sample(posts[:2], tda.CrankNicolson(), 5, subchain_length=3, randomize_subchain_length=True, store_coarse_chain=False)
    with pytest.raises(ValueError):
        tda.sample(posts, tda.CrankNicolson(), 5, subchain_length=[3])
    with pytest.raises(ValueError):
        tda.sample(posts[:2], tda.CrankNicolson(), 5, adaptive_error_model="sometimes")
    with pytest.warns(UserWarning):  # deprecated alias still accepted (sampler.py:113-115)
        with pytest.raises(tda.EngineError):
            tda.sample(posts[:2], tda.CrankNicolson(), 5, subsampling_rate=10)


def test_multilevel_plan_for_plain_callables_and_external_models():
    """Lowering rules that need no GPU: plain Python callables in a hierarchy are wrapped behind the batched-callback
    interface; hierarchies may mix callback, source-defined and linear levels; AdaptiveGaussianLogLike below the finest level."""
    from tinyda_amd import api

    d, m = 4, 9
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, d))
    y = rng.standard_normal(m)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    cov = 0.1 * np.eye(m)
    plain = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), lambda th: np.tanh(A @ th)),
             tda.Posterior(prior, tda.GaussianLogLike(y, cov), lambda th: np.tanh(A @ th) + 0.01)]
    assert api._device_plan(plain, tda.GaussianRandomWalk(np.eye(d))) is None
    wrapped = api._wrap_opaque_models(plain)
    assert wrapped is not None and all(isinstance(p.model, tda.BatchedModel) for p in wrapped)
    out = wrapped[0].model.batch(np.zeros((3, d)))
    assert out.shape == (3, m) and np.allclose(out, 0.0)
    plan = api._device_plan(wrapped, tda.GaussianRandomWalk(np.eye(d), adaptive=True))
    assert plan is not None and "batched" in plan[0][0] and plan[0][0]["noise_kind"] == 3
    mixed = [tda.Posterior(prior, tda.GaussianLogLike(y, cov), tda.LinearModel(A)), wrapped[1]]
    assert api._device_plan(mixed, tda.CrankNicolson(scaling=0.1)) is not None
    assert api._device_plan(mixed, tda.DREAMZ(10)) is not None  # DREAMZ below a hierarchy: host-sequenced, per-chain archives
    assert api._device_plan(mixed, tda.DREAM(10)) is None  # DREAM's shared archive is single-level
    qoi = [tda.Posterior(prior, tda.GaussianLogLike(y, cov), lambda th: (A @ th, th.sum())), wrapped[1]]
    w2 = api._wrap_opaque_models(qoi)  # (output, qoi) models (posterior.py:97-101): the engine takes the output ...
    np.testing.assert_allclose(w2[0].model.batch(np.ones((2, d))), np.tile(A @ np.ones(d), (2, 1)))
    out, q = w2[0].model(np.ones(d))  # ... and one parameter vector still gives the user's own return value
    assert q == d and out.shape == (m,)
    from tinyda_amd.records import DeviceChain

    ln = DeviceChain(np.ones((2, d)), np.zeros((2, 3)), np.ones(2, dtype=np.uint8), w2[0].model)[1]
    assert ln.qoi == d and ln.model_output.shape == (m,)
    bad = api._wrap_opaque_models([tda.Posterior(prior, tda.GaussianLogLike(y, cov), lambda th: [1.0] * m), wrapped[1]])
    with pytest.raises(TypeError):
        bad[0].model.batch(np.zeros((2, d)))


def test_host_hierarchy_fallback_runs_the_references_mlda_notebook_shape():
    """examples/Multilevel Delayed Acceptance.ipynb cells 4-23 in miniature: three levels of an opaque model returning
    (output, qoi), DREAMZ(Z_method='lhs', adaptive=True) at the base, one subchain length for all levels, an explicit start."""
    d = 3
    rng = np.random.default_rng(5)
    truth = np.array([0.3, -0.2, 0.5])
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    posts = []
    for k, m in enumerate((4, 6, 9)):
        B = rng.standard_normal((m, d))
        y = np.tanh(B @ truth) + 0.05 * rng.standard_normal(m)
        posts.append(tda.Posterior(prior, tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(m)), lambda th, B=B: (np.tanh(B @ th), True)))
    np.random.seed(3)
    res = tda.sample(posts, tda.DREAMZ(M0=30, delta=1, Z_method="lhs", adaptive=True, period=5), iterations=12, n_chains=2,
                     initial_parameters=truth, subchain_length=2, backend="host")
    assert res["sampler"] == "MLDA" and res["levels"] == 3 and res["iterations"] == 13 and res["backend"] == "host"
    assert res["subchain_lengths"] == [2, 2]
    assert len(res["chain_l2_0"]) == 13 and len(res["chain_l1_1"]) == 24 and len(res["chain_l0_0"]) == 48
    assert all(ln.qoi is True for ln in res["chain_l2_0"])
    fine = tda.get_samples(res, level=2)
    assert fine["chain_0"].shape == (13, d)
    # Delayed Acceptance with the state-dependent error model, MALA-free non-symmetric proposal (pCN): host protocol
    two = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(posts[1].likelihood.data, 0.05 ** 2 * np.eye(6)), lambda th: np.tanh(th.sum()) * np.ones(6)),
           tda.Posterior(prior, tda.GaussianLogLike(posts[1].likelihood.data, 0.05 ** 2 * np.eye(6)), lambda th: np.tanh(th.sum()) * np.ones(6) + 0.01)]
    res2 = tda.sample(two, tda.CrankNicolson(scaling=0.2), 10, n_chains=1, subchain_length=3, adaptive_error_model="state-dependent",
                      store_coarse_chain=False, backend="host")
    assert res2["sampler"] == "DA" and res2["chain_coarse_0"] is None and len(res2["chain_fine_0"]) == 11


def test_record_buffers_fall_back_when_the_device_is_full(monkeypatch):
    """device records that do not fit: the library's buffer pool is handed back, torch's cache emptied, then page-locked host memory
     -- driven here with a stand-in allocator, no GPU"""
    import types
    import warnings

    from tinyda_amd import api

    calls, released = [], []

    class OOM(RuntimeError):
        pass

    def empty(shape, dtype=None, device=None, pin_memory=False):
        calls.append((tuple(shape), device, pin_memory))
        if device != "cpu" and len([c for c in calls if c[1] != "cpu"]) <= fail_first:
            raise OOM("out of memory")
        return ("buf", tuple(shape), device, pin_memory)

    fake = types.SimpleNamespace(empty=empty, cuda=types.SimpleNamespace(OutOfMemoryError=OOM, empty_cache=lambda: released.append("cache")))
    monkeypatch.setattr(api, "release_cached_memory", lambda: released.append("pool") or 0)
    shapes = [((5, 4, 3), "f64"), ((5, 4), "u8")]
    fail_first = 0
    assert [b[2] for b in api._record_buffers(fake, "cuda:0", shapes)] == ["cuda:0", "cuda:0"] and not released
    calls.clear()
    fail_first = 1  # the first device allocation fails, the retry after the release succeeds
    assert [b[2] for b in api._record_buffers(fake, "cuda:0", shapes)] == ["cuda:0", "cuda:0"] and released == ["pool", "cache"]
    calls.clear()
    del released[:]
    fail_first = 10 ** 6
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        bufs = api._record_buffers(fake, "cuda:0", shapes)
    assert [(b[2], b[3]) for b in bufs] == [("cpu", True), ("cpu", True)] and any(issubclass(x.category, ResourceWarning) for x in w)


def test_result_views_carry_no_instance_dictionary_and_survive_pickle_and_copy():
    """a device result holds one DeviceChain per chain and level (12 288 at BASELINE config 5): slots only -- the interpreter's
    cyclic collector walks every tracked object -- and still picklable / copyable / sliceable like the list of links it stands for"""
    import copy
    import pickle

    from tinyda_amd.records import DeviceChain

    c = DeviceChain(np.arange(15.0).reshape(5, 3), np.zeros((5, 3)), np.ones(5, dtype=np.uint8), None)
    assert not hasattr(c, "__dict__")
    for other in (pickle.loads(pickle.dumps(c)), copy.deepcopy(c), copy.copy(c)):
        assert len(other) == 5 and np.array_equal(other.parameters, c.parameters)
    assert c[1:3].parameters.shape == (2, 3) and c[4].parameters.shape == (3,)


def test_lowering_rules_of_round_5():
    """what _device_plan admits since 0.5 (no GPU needed: the pass only inspects the problem): 65 .. 128 parameters for linear,
    batched and source-defined models under GRW / pCN / AM with any prior kind and the dense error model; five and six levels without
    error model; 256 outputs under the dense error model -- and what it still refuses, with the reason recorded"""
    from tinyda_amd import api

    rng = np.random.default_rng(0)
    d = 100
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    Lp = np.eye(d) + 0.05 * np.tril(rng.standard_normal((d, d)), -1)
    dense_prior = stats.multivariate_normal(np.zeros(d), Lp @ Lp.T)
    A = rng.standard_normal((30, d))
    like = tda.GaussianLogLike(np.zeros(30), 0.1 * np.eye(30))
    lin = tda.Posterior(prior, like, tda.LinearModel(A))
    bat = tda.Posterior(prior, like, tda.BatchedModel(lambda th: th @ A.T, 30))
    am = tda.AdaptiveMetropolis(1e-3 * np.eye(d))
    assert api._device_plan([lin], am) is not None
    assert api._device_plan([tda.Posterior(dense_prior, like, tda.LinearModel(A))], am) is not None
    assert api._device_plan([bat], tda.CrankNicolson()) is not None
    assert api._device_plan([bat, lin], tda.CrankNicolson()) is not None  # a host-sequenced hierarchy at 100 parameters
    dense_like = tda.GaussianLogLike(np.zeros(30), 0.1 * np.eye(30) + 0.01 * np.ones((30, 30)))
    assert api._device_plan([tda.Posterior(prior, dense_like, tda.LinearModel(A))], am) is not None  # dense noise: single level
    assert api._device_plan([lin, tda.Posterior(prior, dense_like, tda.LinearModel(A))], am) is None and "more than 64 parameters" in api._refusal[0]
    ada = tda.Posterior(prior, tda.AdaptiveGaussianLogLike(np.zeros(30), 0.1 * np.eye(30)), tda.LinearModel(0.9 * A))
    assert api._device_plan([ada, lin], tda.CrankNicolson(), error_model="state-independent") is not None
    assert api._device_plan([ada, lin], tda.CrankNicolson(), diagonal_error_model=True, error_model="state-independent") is None
    assert api._device_plan([lin], tda.DREAMZ(M0=200)) is None and "more than 64 parameters" in api._refusal[0]
    assert api._device_plan([tda.Posterior(stats.multivariate_normal(np.zeros(129), np.eye(129)), like, tda.LinearModel(rng.standard_normal((30, 129))))], am) is None
    # levels
    d4 = 4
    prior4 = stats.multivariate_normal(np.zeros(d4), np.eye(d4))
    lv = [tda.Posterior(prior4, tda.GaussianLogLike(np.zeros(6 + k), 0.1 * np.eye(6 + k)), tda.LinearModel(rng.standard_normal((6 + k, d4)))) for k in range(7)]
    grw = tda.GaussianRandomWalk(np.eye(d4))
    assert api._device_plan(lv[:5], grw) is not None and api._device_plan(lv[:6], grw) is not None
    assert api._device_plan(lv[:7], grw) is None and "more than 6 levels" in api._refusal[0]
    same = [tda.Posterior(prior4, tda.AdaptiveGaussianLogLike(np.zeros(6), 0.1 * np.eye(6)), tda.LinearModel(rng.standard_normal((6, d4)))) for _ in range(4)]
    same.append(tda.Posterior(prior4, tda.GaussianLogLike(np.zeros(6), 0.1 * np.eye(6)), tda.LinearModel(rng.standard_normal((6, d4)))))
    assert api._device_plan(same, grw, error_model="state-independent") is None and "more than 4 levels" in api._refusal[0]
    assert api._device_plan(same[1:], grw, error_model="state-independent") is not None
    assert api._device_plan(lv[:5], tda.DREAMZ(M0=20)) is None
