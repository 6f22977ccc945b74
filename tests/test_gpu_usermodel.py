"""Source-defined (hiprtc-compiled) forward models: the fused device path against the oracle running the same model
as a NumPy callable on the exported Philox stream."""
import numpy as np
import pytest
import scipy.stats as st

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu

SRC = """
__device__ double tda_forward(const double* theta, int dim, int o) {
  // output o: sin of a weighted sum plus a quadratic coupling of two parameters
  double s = 0.0;
  for (int j = 0; j < dim; ++j) s += (0.1 + 0.01 * ((o * 7 + j * 3) % 11)) * theta[j];
  const double q = theta[o % dim] * theta[(o + 1) % dim];
  return sin(s) + 0.5 * q;
}
"""


def np_model(theta):
    theta = np.atleast_2d(theta)
    N, d = theta.shape
    m = 23
    out = np.empty((N, m))
    for o in range(m):
        s = np.zeros(N)
        for j in range(d):  # same summation order as the device code
            s = s + (0.1 + 0.01 * ((o * 7 + j * 3) % 11)) * theta[:, j]
        out[:, o] = np.sin(s) + 0.5 * (theta[:, o % d] * theta[:, (o + 1) % d])
    return out


@pytest.mark.parametrize("kind", ["am", "pcn", "grw_diag"])
def test_source_model_matches_oracle(kind):
    from tinyda_amd.engine import Engine

    d, m, N, T = 5, 23, 19, 140
    rng = np.random.default_rng(4)
    truth = 0.5 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    pm, pv = (np.zeros(d), np.ones(d)) if kind == "pcn" else (0.1 * np.arange(d), 0.5 + 0.1 * np.arange(d))
    e = Engine(N, d, seed=77, chain_offset=2, block_steps=33)
    e.set_prior(pm, np.diag(pv))
    noise = 0.05 ** 2 * (1.0 + 0.1 * np.arange(m))
    if kind == "grw_diag":
        e.set_level_source(0, SRC, y, 1, noise)
        lvl = orc.CallableGaussianLevel(np_model, y, "diag", noise, orc.MVNPrior(pm, np.diag(pv)))
        e.set_proposal(0, 2e-3 * np.eye(d), scaling=1.0, adaptive=True, period=20)
        prop = dict(kind="grw", C=2e-3 * np.eye(d), scaling=1.0, adaptive=True, period=20)
    else:
        e.set_level_source(0, SRC, y, 0, [0.05 ** 2])
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
    ref = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    assert 0.05 < acc.mean() < 0.95


def test_source_model_through_sample_api():
    import tinyda_amd as tda

    d, m = 5, 23
    rng = np.random.default_rng(1)
    truth = 0.3 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(m)
    post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(m)),
                         tda.DeviceModel(SRC, m, reference=lambda t: np_model(t)[0]))
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d), t0=50, period=50), 300, n_chains=8,
                     initial_parameters=truth, seed=3)
    assert res["sampler"] == "MH" and res["n_chains"] == 8
    link = res["chain_0"][-1]
    assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)


def test_source_that_does_not_compile_is_reported():
    from tinyda_amd import _lib
    from tinyda_amd.engine import Engine

    e = Engine(4, 3, seed=1)
    e.set_prior(np.zeros(3), np.eye(3))
    with pytest.raises(_lib.EngineError, match="does not compile"):
        e.set_level_source(0, "__device__ double tda_forward(const double* t, int d, int o) { return undefined_symbol; }",
                           np.zeros(2), 0, [1.0])
    e.close()


def test_source_model_with_joint_prior_bounds():
    """A source-defined model under a JointPrior with uniform components: proposals outside the support are rejected, the
    trace equals the oracle's."""
    from tinyda_amd.engine import Engine

    d, m, N, T = 5, 23, 12, 120
    rng = np.random.default_rng(8)
    kinds = np.array([1, 0, 1, 0, 1])
    loc = np.array([-0.4, 0.0, -0.3, 0.1, -0.5])
    scale = np.array([0.8, 0.7, 0.6, 0.5, 1.0])
    truth = np.array([0.0, 0.2, 0.0, -0.1, 0.1])
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    e = Engine(N, d, seed=5)
    e.set_prior_joint(kinds, loc, scale)
    e.set_level_source(0, SRC, y, 0, [0.05 ** 2])
    e.set_proposal(0, 0.02 * np.eye(d), scaling=1.0, adaptive=True, period=20)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    e.close()
    lvl = orc.CallableGaussianLevel(np_model, y, "iso", 0.05 ** 2, orc.JointPriorOracle(kinds, loc, scale))
    ref = orc.run_mh(lvl, dict(kind="grw", C=0.02 * np.eye(d), scaling=1.0, adaptive=True, period=20), theta0,
                     np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    lo, hi = np.where(kinds == 1, loc, -np.inf), np.where(kinds == 1, loc + scale, np.inf)
    assert np.all(params >= lo) and np.all(params <= hi)
    assert (acc == 0).sum() > 0


SRC_LEVEL = """
__device__ double tda_forward(const double* theta, int dim, int o) {
  // fidelity %(k)d of the model above: coarser levels weaken the quadratic coupling and shift the weights
  double s = 0.0;
  for (int j = 0; j < dim; ++j) s += (0.1 + 0.01 * ((o * 7 + j * 3) %% 11) + %(shift).4f) * theta[j];
  const double q = theta[o %% dim] * theta[(o + 1) %% dim];
  return sin(s) + %(coup).4f * q;
}
"""


def _level_cfg(k):
    return dict(k=k, shift=0.004 * (2 - k), coup=0.5 * (0.6 + 0.2 * k))


def np_level_model(k):
    cfg = _level_cfg(k)

    def fn(theta):
        theta = np.atleast_2d(theta)
        N, d = theta.shape
        m = 23
        out = np.empty((N, m))
        for o in range(m):
            s = np.zeros(N)
            for j in range(d):
                s = s + (0.1 + 0.01 * ((o * 7 + j * 3) % 11) + float("%.4f" % cfg["shift"])) * theta[:, j]
            out[:, o] = np.sin(s) + float("%.4f" % cfg["coup"]) * (theta[:, o % d] * theta[:, (o + 1) % d])
        return out
    return fn


@pytest.mark.parametrize("case", ["da_pcn", "mlda_am", "da_mixed", "da_linear_coarse", "da_random"])
def test_source_model_hierarchy_matches_oracle(case):
    """Delayed Acceptance / MLDA with source-defined (hiprtc) models at every level -- no host round trip per step -- and a
    hierarchy mixing a batched host callback (coarse) with a source-defined fine model, against the oracle's DAChain /
    MLDAChain restatement running the NumPy twins of the models."""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, m, N = 5, 23, 21
    rng = np.random.default_rng(8)
    truth = 0.5 * rng.standard_normal(d)
    if case == "mlda_am":
        ks, sl, n_fine, block = [0, 1, 2], [3, 2], 14, 7
    else:
        ks, sl, n_fine, block = [1, 2], [4], 25, 0
    nl = len(ks)
    twins = [np_level_model(k) for k in ks]
    y = twins[-1](truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    seed = 991
    e = Engine(N, d, seed=seed, n_levels=nl, block_steps=block)
    e.set_prior(pm, np.diag(pv))
    if case == "da_linear_coarse":  # a linear surrogate (the model linearised at the origin) below the non-linear model
        Alin = np.array([[(0.1 + 0.01 * ((o * 7 + j * 3) % 11)) for j in range(d)] for o in range(m)])
        twins[0] = lambda th, A=Alin: np.atleast_2d(th) @ A.T
    for i, k in enumerate(ks):
        if case == "da_linear_coarse" and i == 0:
            e.set_level(0, Alin, y, 0, 0.25 ** 2)  # inflated variance for the crude surrogate
        elif case == "da_mixed" and i == 0:
            e.set_level_callback(0, twins[0], y, 0, [0.05 ** 2])
        else:
            e.set_level_source(i, SRC_LEVEL % _level_cfg(k), y, 0, [0.05 ** 2])
    if case == "mlda_am":
        e.set_proposal(2, 2e-3 * np.eye(d), t0=20, period=10, adaptive=True, gamma=1.02)
        prop = dict(kind="am", C0=2e-3 * np.eye(d), t0=20, period=10, adaptive=True, gamma=1.02)
    else:  # adaptive scaling: the accept-flag window holds the base steps of the fused launches and the alignment entries
        e.set_proposal(1, None, scaling=0.04, adaptive=True, gamma=1.02, period=15)
        prop = dict(kind="pcn", scaling=0.04, adaptive=True, gamma=1.02, period=15)
    e.set_subchains(sl, case == "da_random")  # randomize_subchain_length: a random state of the subchain is promoted
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    scal = e.proposal_state()["scaling"]
    e.close()
    us, ridx = _oracle_uniforms(seed, N, rows, sl, sl[0] if case == "da_random" else None)
    prior = orc.MVNPrior(pm, np.diag(pv))
    nvar = [0.25 ** 2 if (case == "da_linear_coarse" and i == 0) else 0.05 ** 2 for i in range(nl)]
    levels = [orc.CallableGaussianLevel(twins[i], y, "iso", nvar[i], prior) for i in range(nl)]
    res, pstate = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, ridx)
    np.testing.assert_allclose(scal, pstate.scaling, rtol=1e-12)
    for i in range(nl):
        ref = res[i]
        sk = slice(1, None) if i == nl - 1 else slice(None)
        assert np.array_equal(outs[i][2], ref["accepted"][:, sk].T), "level %d accept masks differ" % i
        np.testing.assert_allclose(outs[i][1][:, :, 2], ref["logpost"][:, sk].T, rtol=1e-10)
    assert 0.02 < outs[nl - 1][2].mean() < 0.98


def test_source_model_hierarchy_through_sample_api():
    import tinyda_amd as tda

    d, m = 5, 23
    rng = np.random.default_rng(12)
    truth = 0.4 * rng.standard_normal(d)
    y = np_level_model(2)(truth)[0] + 0.05 * rng.standard_normal(m)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    like = tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(m))
    posts = [tda.Posterior(prior, like, tda.DeviceModel(SRC_LEVEL % _level_cfg(k), m, reference=lambda th, k=k: np_level_model(k)(th)[0]))
             for k in (1, 2)]
    th0 = [truth + 0.05 * rng.standard_normal(d) for _ in range(8)]
    res = tda.sample(posts, tda.GaussianRandomWalk(2e-3 * np.eye(d)), 30, n_chains=8, initial_parameters=th0, subchain_length=3, seed=5)
    assert res["sampler"] == "DA" and res.get("backend", "hip") != "host"
    link = res["chain_fine_5"][-1]
    assert np.isclose(link.posterior, posts[1].create_link(link.parameters).posterior, rtol=1e-10)


@pytest.mark.parametrize("case", ["da_source", "mlda_mixed", "da_dep_pcn", "da_dep_grw"])
def test_hierarchy_with_error_model_matches_oracle(case):
    """State-independent adaptive error model over non-linear models: source-defined levels (DA), and a 3-level MLDA
    hierarchy of a linear surrogate, a batched host callback and a source-defined finest level -- against the oracle's
    restatement of the reference's error-model chains running the NumPy twins."""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, m, N = 5, 23, 17
    rng = np.random.default_rng(21)
    truth = 0.5 * rng.standard_normal(d)
    if case == "mlda_mixed":
        ks, sl, n_fine = [0, 1, 2], [3, 2], 12
    elif case.startswith("da_dep"):
        ks, sl, n_fine = [1, 2], [1 if case == "da_dep_pcn" else 2], 40
    else:
        ks, sl, n_fine = [1, 2], [3], 22
    aem = "state-dependent" if case.startswith("da_dep") else "state-independent"
    nl = len(ks)
    twins = [np_level_model(k) for k in ks]
    Alin = np.array([[(0.1 + 0.01 * ((o * 7 + j * 3) % 11)) for j in range(d)] for o in range(m)])
    blin = 0.01 * np.arange(m)
    if case == "mlda_mixed":
        twins[0] = lambda th: np.atleast_2d(th) @ Alin.T + blin
    y = twins[-1](truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    var = 0.05 ** 2
    cov = var * np.eye(m)
    seed = 313
    e = Engine(N, d, seed=seed, n_levels=nl)
    e.set_prior(pm, np.diag(pv))
    for i, k in enumerate(ks):
        last = i == nl - 1
        if case == "mlda_mixed" and i == 0:
            e.set_level(0, Alin, y, 3, cov, b=blin)
        elif case == "mlda_mixed" and i == 1:
            e.set_level_callback(1, twins[1], y, 3, cov)
        else:
            e.set_level_source(i, SRC_LEVEL % _level_cfg(k), y, 0 if last else 3, [var] if last else cov)
    if case == "da_dep_pcn":
        e.set_proposal(1, None, scaling=0.05)
        prop = dict(kind="pcn", scaling=0.05)
    else:
        e.set_proposal(0, 2e-3 * np.eye(d), scaling=1.0)
        prop = dict(kind="grw", C=2e-3 * np.eye(d), scaling=1.0)
    e.set_subchains(sl, False)
    e.set_error_model(aem)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    _, st_init = e.level_state(nl - 1)
    h1 = n_fine // 2
    o1 = e.run_levels_host(h1)
    o2 = e.run_levels_host(n_fine - h1)
    outs = [tuple(np.concatenate([x1, x2]) for x1, x2 in zip(a1, a2)) for a1, a2 in zip(o1, o2)]
    bias, P = e.error_model_state(0, m)
    e.close()
    us, _ = _oracle_uniforms(seed, N, rows, sl)
    prior = orc.MVNPrior(pm, np.diag(pv))
    levels = [dict(fn=twins[i], y=y, prior=prior, **(dict(var=var) if i == nl - 1 else dict(cov=cov))) for i in range(nl)]
    res = orc.run_multilevel_aem(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, aem)
    res = res[0] if isinstance(res, tuple) else res
    for i in range(nl):
        ref = res[i]
        sk = slice(1, None) if i == nl - 1 else slice(None)
        assert np.array_equal(outs[i][2], np.asarray(ref["accepted"])[:, sk].T), "level %d accept masks differ" % i
        np.testing.assert_allclose(outs[i][0], np.swapaxes(np.asarray(ref["theta"])[:, sk], 0, 1), rtol=1e-9, atol=1e-11)
        if i == nl - 1:  # coarse links are refreshed after they were recorded (update_link), the finest never is
            np.testing.assert_allclose(outs[i][1][:, :, 2], (np.asarray(ref["logprior"]) + np.asarray(ref["loglike"]))[:, sk].T, rtol=1e-10)
    assert 0.05 < outs[nl - 1][2].mean() < 0.98 and np.all(np.isfinite(bias)) and np.all(np.isfinite(P))


def test_sample_api_hierarchy_with_error_model():
    """tda.sample([...], adaptive_error_model='state-independent') with AdaptiveGaussianLogLike over a batched host model
    (coarse) and a source-defined model (fine)."""
    import tinyda_amd as tda

    d, m = 5, 23
    rng = np.random.default_rng(31)
    truth = 0.4 * rng.standard_normal(d)
    y = np_level_model(2)(truth)[0] + 0.05 * rng.standard_normal(m)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    cov = 0.05 ** 2 * np.eye(m)
    posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), tda.BatchedModel(np_level_model(1), m)),
             tda.Posterior(prior, tda.GaussianLogLike(y, cov),
                           tda.DeviceModel(SRC_LEVEL % _level_cfg(2), m, reference=lambda th: np_level_model(2)(th)[0]))]
    th0 = [truth + 0.05 * rng.standard_normal(d) for _ in range(8)]
    res = tda.sample(posts, tda.GaussianRandomWalk(2e-3 * np.eye(d)), 25, n_chains=8, initial_parameters=th0, subchain_length=3,
                     adaptive_error_model="state-independent", seed=5)
    assert res["sampler"] == "DA" and res.get("backend", "hip") != "host"
    link = res["chain_fine_3"][-1]
    assert np.isclose(link.posterior, posts[1].create_link(link.parameters).posterior, rtol=1e-10)
    assert np.mean([np.mean(res["chain_fine_%d" % i].accepted[1:]) for i in range(8)]) > 0.05


def test_hierarchy_with_error_model_checkpoint_resume():
    """get_state / set_state of a source-defined DA hierarchy with the error model: stored model outputs, trackers and
    inverses come back, the continuation is bitwise the same."""
    from tinyda_amd.engine import Engine

    d, m, N = 5, 23, 12
    rng = np.random.default_rng(41)
    truth = 0.5 * rng.standard_normal(d)
    y = np_level_model(2)(truth)[0] + 0.05 * rng.standard_normal(m)
    e = Engine(N, d, seed=77, n_levels=2)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_source(0, SRC_LEVEL % _level_cfg(1), y, 3, 0.05 ** 2 * np.eye(m))
    e.set_level_source(1, SRC_LEVEL % _level_cfg(2), y, 0, [0.05 ** 2])
    e.set_proposal(0, 2e-3 * np.eye(d), scaling=1.0, adaptive=True, period=10)
    e.set_subchains([3], False)
    e.set_error_model("state-independent")
    e.init(truth + 0.05 * rng.standard_normal((N, d)))
    e.run_levels_host(7)
    blob = e.get_state()
    a = e.run_levels_host(9)
    e.set_state(blob)
    b = e.run_levels_host(9)
    e.close()
    for la, lb in zip(a, b):
        assert all(np.array_equal(x, y_) for x, y_ in zip(la, lb))


def test_hierarchy_with_uniform_prior_components():
    """JointPrior of uniform and normal components over a Delayed-Acceptance pair of source-defined models, with the error
    model: proposals outside a uniform support are rejected by the base-level kernels; against the oracle."""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, m, N, sl, n_fine = 5, 23, 17, [3], 24
    rng = np.random.default_rng(51)
    kinds = np.array([1, 0, 1, 0, 0])
    loc = np.array([-0.6, 0.0, -0.5, 0.1, 0.0])
    scale = np.array([1.2, 1.0, 1.0, 0.8, 1.0])  # uniform on [loc, loc + scale]
    truth = np.array([0.45, 0.3, 0.42, -0.2, 0.1])  # two components close to the upper edge of their support
    twins = [np_level_model(1), np_level_model(2)]
    y = twins[1](truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.03 * rng.standard_normal((N, d))
    var, cov, seed = 0.05 ** 2, 0.05 ** 2 * np.eye(m), 515
    e = Engine(N, d, seed=seed, n_levels=2)
    e.set_prior_joint(kinds, loc, scale)
    e.set_level_source(0, SRC_LEVEL % _level_cfg(1), y, 3, cov)
    e.set_level_source(1, SRC_LEVEL % _level_cfg(2), y, 0, [var])
    e.set_proposal(0, 6e-3 * np.eye(d), scaling=1.0)
    e.set_subchains(sl, False)
    e.set_error_model("state-independent")
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    e.close()
    us, _ = _oracle_uniforms(seed, N, rows, sl)
    prior = orc.JointPriorOracle(kinds, loc, scale)
    levels = [dict(fn=twins[0], y=y, prior=prior, cov=cov), dict(fn=twins[1], y=y, prior=prior, var=var)]
    res = orc.run_multilevel_aem(levels, dict(kind="grw", C=6e-3 * np.eye(d), scaling=1.0), sl, theta0, np.swapaxes(z, 0, 1), us, n_fine,
                                 "state-independent")
    res = res[0] if isinstance(res, tuple) else res
    for i in range(2):
        sk = slice(1, None) if i == 1 else slice(None)
        assert np.array_equal(outs[i][2], np.asarray(res[i]["accepted"])[:, sk].T), "level %d accept masks differ" % i
        np.testing.assert_allclose(outs[i][0], np.swapaxes(np.asarray(res[i]["theta"])[:, sk], 0, 1), rtol=1e-9, atol=1e-11)
    th = outs[1][0]
    assert (th[:, :, 0] <= loc[0] + scale[0]).all() and (th[:, :, 2] <= loc[2] + scale[2]).all() and (th[:, :, 0] >= loc[0]).all()
    assert 0.05 < outs[1][2].mean() < 0.98
