"""Single-level chains with 65 .. 128 parameters (round 5, tda_kernels_wide.h: k_mh_steps<128, 4>, k_rng<128>, k_wide_adapt, the error
model's blocked Cholesky as the covariance swap, k_wide_apply) against the oracle on the engine's own Philox stream.  Same bar as
test_gpu_parity.py: accept masks bit-exact, log-posterior within 1e-10 relative."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu


def _spd(rng, n, scale):
    B = rng.standard_normal((n, n))
    return scale * (np.eye(n) + 0.3 * B @ B.T / n)


def _problem(d, m, N, seed, noise="iso"):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = 0.7 * rng.standard_normal(d)
    y = A @ truth + 0.1 * rng.standard_normal(m)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    nz = 0.01 if noise == "iso" else 0.01 * (0.5 + rng.random(m))
    return rng, A, y, theta0, nz


CASES = [
    # (d, m, N, T, proposal, noise, prior)
    (96, 40, 17, 130, "am", "iso", "identity"),
    (128, 64, 16, 90, "am_adaptive", "diag", "diag"),
    (65, 17, 5, 75, "am", "iso", "diag"),
    (100, 130, 33, 60, "grw_adaptive", "iso", "identity"),
    (128, 33, 16, 50, "pcn_adaptive", "diag", "identity"),
    (80, 257, 20, 45, "pcn", "iso", "identity"),
]


@pytest.mark.parametrize("case", CASES, ids=["%d-%s" % (c[0], c[4]) for c in CASES])
def test_wide_single_level_against_the_oracle(case):
    from tinyda_amd.engine import Engine

    d, m, N, T, kind, noise, prior = case
    rng, A, y, theta0, nz = _problem(d, m, N, 100 + d + m, noise)
    if prior == "identity":
        pm, pc = np.zeros(d), np.eye(d)
    else:
        pm, pc = (np.zeros(d) if kind.startswith("pcn") else 0.1 * rng.standard_normal(d)), np.diag(0.5 + rng.random(d))
    period = 20
    C0 = _spd(rng, d, 8e-3 / d)
    adaptive = kind.endswith("adaptive")
    e = Engine(N, d, seed=77 + d, chain_offset=3, block_steps=33 if d == 100 else 0)
    e.set_prior(pm, pc)
    e.set_level(0, A, y, 0 if noise == "iso" else 1, nz)
    if kind.startswith("am"):
        e.set_proposal(2, C0, t0=period, period=period, adaptive=adaptive)
        prop = dict(kind="am", C0=C0, t0=period, period=period, adaptive=adaptive)
    elif kind.startswith("grw"):
        e.set_proposal(0, C0, scaling=0.8, adaptive=adaptive, period=period, gamma=1.05)
        prop = dict(kind="grw", C=C0, scaling=0.8, adaptive=adaptive, period=period, gamma=1.05)
    else:
        e.set_proposal(1, None, scaling=0.03, adaptive=adaptive, period=period)
        prop = dict(kind="pcn", scaling=0.03, adaptive=adaptive, period=period)
    e.init(theta0)
    z, u = e.set_export(T)
    k = T // 3  # two run() calls continue the same chains
    p1, s1, a1 = e.run_host(k)
    p2, s2, a2 = e.run_host(T - k)
    params, stats, acc = np.concatenate([p1, p2]), np.concatenate([s1, s2]), np.concatenate([a1, a2])
    st = e.proposal_state(want_am=kind.startswith("am"))
    e.close()
    lvl = orc.LinearGaussianLevel(A, y, noise, nz, orc.MVNPrior(pm, pc))
    res = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    ref_acc = np.swapaxes(res["accepted"][:, 1:], 0, 1)
    assert np.array_equal(acc, ref_acc), "%d accept flips" % int((acc != ref_acc).sum())
    assert 0.02 < acc.mean() < 0.98
    # (AdaptiveMetropolis with fewer states than 4 d behind a swap: the near-singular factor carries the recursion's last bits into the
    # proposals -- the rule of tests/test_gpu_sweep.py)
    rtol = 1e-9 if (kind.startswith("am") and T < 4 * d) else 1e-10
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(res["logpost"][:, 1:], 0, 1), rtol=rtol)
    np.testing.assert_allclose(params, np.swapaxes(res["theta"][:, 1:], 0, 1), rtol=1e-8, atol=1e-10)
    if kind.startswith("am"):
        np.testing.assert_allclose(st["am_sigma"], res["am_sigma"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(st["am_mu"], res["am_mu"], rtol=1e-8, atol=1e-10)  # (means of states that agree to 1e-8 / 1e-10)
    if adaptive:
        np.testing.assert_allclose(st["scaling"], res["scaling"], rtol=1e-12)


def test_sample_runs_100_parameters_on_the_device():
    """tda.sample() at 100 parameters returns "backend": "hip" (0.4: the host protocol, with a HostFallbackWarning) and a chain that has
    moved towards the data; Delayed Acceptance and three-level MLDA at 100 parameters run there too; the dense error model too; the diagonal error model above 64
    parameters still falls back, announced"""
    import warnings

    import scipy.stats as stats

    import tinyda_amd as tda

    d, m, n_chains, T = 100, 300, 32, 400
    rng = np.random.default_rng(9)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = 0.5 * rng.standard_normal(d)
    y = A @ truth + 0.05 * rng.standard_normal(m)
    post = tda.Posterior(stats.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.0025 * np.eye(m)), tda.LinearModel(A))
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=n_chains, seed=5)
    assert res["backend"] == "hip" and res["iterations"] == T + 1
    par = tda.get_samples(res, burnin=T // 2)["chain_3"]
    assert par.shape == (T + 1 - T // 2, d)
    first, last = res["chain_0"][0], res["chain_0"][-1]
    assert last.posterior > first.posterior + 100.0  # (theta0 ~ prior: far from the data)
    st = res["proposal_state"]
    assert np.asarray(st["C"]).shape == (n_chains, d, d) and np.all(np.isfinite(np.asarray(st["C"])))
    coarse = tda.Posterior(post.prior, tda.GaussianLogLike(y[:50], 0.0025 * np.eye(50)), tda.LinearModel(A[:50]))
    da = tda.sample([coarse, post], tda.CrankNicolson(scaling=0.02), 40, n_chains=8, subchain_length=3, seed=3)  # Delayed Acceptance at 100 parameters
    assert da["backend"] == "hip" and da["sampler"] == "DA" and len(da["chain_fine_0"]) == 41
    mid = tda.Posterior(post.prior, tda.GaussianLogLike(y[:120], 0.0025 * np.eye(120)), tda.LinearModel(A[:120]))
    ml = tda.sample([coarse, mid, post], tda.CrankNicolson(scaling=0.02), 12, n_chains=8, subchain_length=[2, 2], seed=4)  # three-level MLDA
    assert ml["backend"] == "hip" and ml["sampler"] == "MLDA"
    Lp = np.eye(d) + 0.05 * np.tril(rng.standard_normal((d, d)), -1)
    dense = tda.Posterior(stats.multivariate_normal(np.zeros(d), Lp @ Lp.T), post.likelihood, post.model)  # a dense prior covariance
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        dn = tda.sample(dense, tda.GaussianRandomWalk(1e-4 * np.eye(d), adaptive=True, period=50), 60, n_chains=8, seed=6)
    assert dn["backend"] == "hip"
    lk = dn["chain_2"][-1]
    rf = dense.create_link(lk.parameters)
    np.testing.assert_allclose([lk.prior, lk.likelihood], [rf.prior, rf.likelihood], rtol=1e-10)
    acoarse = tda.Posterior(post.prior, tda.AdaptiveGaussianLogLike(y[:50], 0.0025 * np.eye(50)), tda.LinearModel(0.9 * A[:50]))  # (an error model: same outputs on both levels)
    with warnings.catch_warnings():  # the dense error model at 100 parameters: on the device (k_ml_steps<128, 1> + k_aem_action<128>)
        warnings.simplefilter("error", tda.HostFallbackWarning)
        em = tda.sample([acoarse, coarse], tda.CrankNicolson(scaling=0.02), 10, n_chains=6, subchain_length=2, adaptive_error_model="state-independent", seed=7)
    assert em["backend"] == "hip" and em["sampler"] == "DA"
    lk = em["chain_fine_1"][-1]
    rf = coarse.create_link(lk.parameters)
    np.testing.assert_allclose([lk.prior, lk.likelihood], [rf.prior, rf.likelihood], rtol=1e-10)
    with pytest.warns(tda.HostFallbackWarning, match="more than 64 parameters are lowered for single-level chains, Delayed Acceptance and MLDA"):
        tda.sample([acoarse, coarse], tda.CrankNicolson(scaling=0.05), 2, n_chains=1, subchain_length=2, adaptive_error_model="state-independent",
                   error_model_covariance="diagonal")


DA_CASES = [(96, (24, 70), 3, 17, 12, "pcn"), (128, (33, 130), 4, 16, 9, "am"), (80, (16, 40), 2, 20, 15, "grw_adaptive"), (100, (20, 65), 3, 18, 10, "pcn_random"),
            (112, (20, 48, 150), (3, 2), 16, 8, "am"), (128, (16, 40, 90, 200), (3, 2, 2), 16, 5, "pcn"), (72, (12, 30, 64), (4, 3), 18, 6, "grw_adaptive")]


@pytest.mark.parametrize("case", DA_CASES, ids=["%d-%dlev-%s" % (c[0], len(c[1]), c[5]) for c in DA_CASES])
def test_wide_delayed_acceptance_against_the_oracle(case):
    """Delayed Acceptance and three- / four-level MLDA at 72 .. 128 parameters (generic level kernel k_ml_steps<128, 2..4> + the wide
    proposal / adaptation launches) on the engine's own Philox stream against the oracle: accept masks of every level exact,
    log-posteriors to 1e-10"""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, ms, L, N, n_fine, kind = case
    sl = list(L) if isinstance(L, tuple) else [L]
    nl = len(ms)
    rng = np.random.default_rng(7000 + d)
    truth = 0.5 * rng.standard_normal(d)
    As = [rng.standard_normal((m, d)) / np.sqrt(d) for m in ms]
    ys = [A @ truth + 0.1 * rng.standard_normal(len(A)) for A in As]
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    period = 8
    C0 = _spd(rng, d, 4e-3 / d)
    randomize = kind.endswith("random")
    if kind.startswith("pcn"):
        prop = dict(kind="pcn", scaling=0.04, adaptive=True, gamma=1.01, period=period)
    elif kind.startswith("grw"):
        prop = dict(kind="grw", C=C0, scaling=0.9, adaptive=True, gamma=1.02, period=period)
    else:
        prop = dict(kind="am", C0=C0, t0=period, period=period, adaptive=False)
    seed = 900 + d
    e = Engine(N, d, seed=seed, n_levels=nl)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(nl):
        e.set_level(k, As[k], ys[k], 0, 0.01)
    if prop["kind"] == "pcn":
        e.set_proposal(1, None, scaling=prop["scaling"], adaptive=True, gamma=prop["gamma"], period=period)
    elif prop["kind"] == "grw":
        e.set_proposal(0, C0, scaling=0.9, adaptive=True, gamma=1.02, period=period)
    else:
        e.set_proposal(2, C0, t0=period, period=period, adaptive=False)
    e.set_subchains(sl, randomize)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    e.close()
    us, ridx = _oracle_uniforms(seed, N, rows, sl, sl[0] if randomize else None)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(As[k], ys[k], "iso", 0.01, prior) for k in range(nl)]
    res, _ = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, ridx)
    for k in range(nl):
        ref = res[k]
        sk = slice(1, None) if k == nl - 1 else slice(None)
        assert np.array_equal(outs[k][2], ref["accepted"][:, sk].T), "level %d accept masks differ" % k
        np.testing.assert_allclose(outs[k][1][:, :, 2], ref["logpost"][:, sk].T, rtol=1e-9 if kind == "am" else 1e-10)
    assert 0.0 < outs[0][2].mean() < 1.0


def test_wide_checkpoint_resume_is_bitwise():
    """get_state / set_state at 96 parameters: the blob carries the factor tiles of both buffers, the diagonal tiles and the selector; a
    run interrupted mid period and resumed in a fresh engine continues bit for bit across two covariance swaps"""
    from tinyda_amd.engine import Engine

    d, m, N = 96, 50, 19
    rng, A, y, theta0, nz = _problem(d, m, N, 4321)
    C0 = _spd(rng, d, 8e-3 / d)

    def make():
        e = Engine(N, d, seed=11, chain_offset=2)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level(0, A, y, 0, nz)
        e.set_proposal(2, C0, t0=20, period=20, adaptive=True)
        e.init(theta0)
        return e

    a = make()
    full = a.run_host(95)
    a.close()
    b = make()
    first = b.run_host(33)
    blob = b.get_state()
    b.close()
    c = make()
    c.set_state(blob)
    rest = c.run_host(62)
    st = c.proposal_state(want_am=True)
    c.close()
    for k in range(3):
        assert np.array_equal(np.concatenate([first[k], rest[k]]), full[k])
    assert np.all(np.isfinite(st["C"])) and st["t"] == 95


@pytest.mark.parametrize("kind", ["am", "pcn", "grw_diag"])
def test_wide_batched_host_model_matches_the_oracle(kind):
    """a batched host callback (tda.BatchedModel / tda_engine_set_level_callback) at 100 parameters, single level: the device draws,
    proposes, decides and adapts, the model is evaluated on the host for all chains at once -- against the oracle running the same
    NumPy model chain by chain on the exported stream"""
    from tinyda_amd.engine import Engine

    d, M, N, T = 100, 70, 19, 90
    W = 0.02 + 0.003 * ((np.arange(M)[:, None] * 7 + np.arange(d)[None, :] * 3) % 11)

    def np_model(theta):
        theta = np.atleast_2d(theta)
        return np.tanh(theta @ W.T) + 0.25 * theta[:, [0]] * theta[:, [-1]]

    rng = np.random.default_rng(15)
    truth = 0.3 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(M)
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), (np.ones(d) if kind == "pcn" else 0.5 + 0.01 * np.arange(d))
    calls = []

    def fn(thetas):
        calls.append(thetas.shape)
        return np_model(thetas)

    e = Engine(N, d, seed=92, chain_offset=1, block_steps=32)
    e.set_prior(pm, np.diag(pv))
    noise = 0.05 ** 2 * (1.0 + 0.1 * np.arange(M))
    C0 = _spd(rng, d, 2e-4 / d)
    if kind == "grw_diag":
        e.set_level_callback(0, fn, y, 1, noise)
        lvl = orc.CallableGaussianLevel(np_model, y, "diag", noise, orc.MVNPrior(pm, np.diag(pv)))
        e.set_proposal(0, C0, scaling=1.0, adaptive=True, period=20)
        prop = dict(kind="grw", C=C0, scaling=1.0, adaptive=True, period=20)
    else:
        e.set_level_callback(0, fn, y, 0, [0.05 ** 2])
        lvl = orc.CallableGaussianLevel(np_model, y, "iso", 0.05 ** 2, orc.MVNPrior(pm, np.diag(pv)))
        if kind == "am":
            e.set_proposal(2, C0, t0=30, period=30)
            prop = dict(kind="am", C0=C0, t0=30, period=30)
        else:
            e.set_proposal(1, None, scaling=0.01)
            prop = dict(kind="pcn", scaling=0.01)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    e.close()
    assert calls == [(N, d)] * (T + 1)
    ref = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-9 if kind == "am" else 1e-10)
    np.testing.assert_allclose(params, np.swapaxes(ref["theta"][:, 1:], 0, 1), rtol=1e-9, atol=1e-12)
    assert 0.0 < acc.mean() < 1.0


def test_wide_batched_model_through_sample():
    """tda.sample with a BatchedModel at 80 parameters returns backend hip (0.4: the host protocol chain by chain), single level and as
    a Delayed-Acceptance hierarchy of two batched models"""
    import warnings

    import scipy.stats as stats

    import tinyda_amd as tda

    d, m, n_chains = 80, 40, 12
    rng = np.random.default_rng(2)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    y = np.tanh(A @ (0.3 * rng.standard_normal(d))) + 0.05 * rng.standard_normal(m)
    model = tda.BatchedModel(lambda th: np.tanh(th @ A.T), m)
    post = tda.Posterior(stats.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.0025 * np.eye(m)), model)
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        res = tda.sample(post, tda.CrankNicolson(scaling=0.05), 60, n_chains=n_chains, seed=1)
    assert res["backend"] == "hip"
    link = res["chain_5"][-1]
    ref = post.create_link(link.parameters)
    np.testing.assert_allclose([link.prior, link.likelihood], [ref.prior, ref.likelihood], rtol=1e-10)
    coarse = tda.Posterior(post.prior, tda.GaussianLogLike(y[:20], 0.0025 * np.eye(20)), tda.BatchedModel(lambda th: np.tanh(th @ A[:20].T), 20))
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        da = tda.sample([coarse, post], tda.CrankNicolson(scaling=0.05), 20, n_chains=n_chains, subchain_length=3, seed=2)
    assert da["backend"] == "hip" and da["sampler"] == "DA"
    lk = da["chain_fine_3"][-1]
    rf = post.create_link(lk.parameters)
    np.testing.assert_allclose([lk.prior, lk.likelihood], [rf.prior, rf.likelihood], rtol=1e-10)


@pytest.mark.parametrize("case", ["da_pcn", "da_grw_random", "mlda_am", "mlda_mixed"])
def test_wide_callback_hierarchy_matches_the_oracle(case):
    """Delayed Acceptance / MLDA at 90 parameters with the levels behind batched host callbacks (host-sequenced level actions:
    k_ext_level_action with a second parameter per lane; mlda_mixed: a LINEAR coarse level beside two callback levels, its outputs
    from k_linear_outputs<128>; da_grw_random: randomised subchain lengths, the promoted state through ysnap) against the oracle
    running the same NumPy models"""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, M, N = 90, 40, 17
    rng = np.random.default_rng(16)
    truth = 0.3 * rng.standard_normal(d)
    Wb = 0.02 + 0.003 * ((np.arange(M)[:, None] * 7 + np.arange(d)[None, :] * 3) % 11)

    def level_model(k):
        def fn(theta):
            theta = np.atleast_2d(theta)
            return np.tanh(theta @ (Wb + 0.001 * (2 - k)).T) + 0.25 * (0.6 + 0.2 * k) * theta[:, [0]] * theta[:, [-1]]
        return fn

    if case.startswith("mlda"):
        nl, sl, n_fine = 3, [3, 2], 10
    else:
        nl, sl, n_fine = 2, [4], 18
    models = [level_model(k + (3 - nl)) for k in range(nl)]
    A0 = rng.standard_normal((M, d)) / np.sqrt(d)
    if case == "mlda_mixed":
        models[0] = lambda th: np.atleast_2d(th) @ A0.T
    y = models[-1](truth)[0] + 0.05 * rng.standard_normal(M)
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    noise = 0.05 ** 2 if case != "mlda_mixed" else 0.3 ** 2
    randomize = case == "da_grw_random"
    seed = 4712
    e = Engine(N, d, seed=seed, n_levels=nl)
    e.set_prior(pm, np.diag(pv))
    for k in range(nl):
        if case == "mlda_mixed" and k == 0:
            e.set_level(0, A0, y, 0, noise)
        else:
            e.set_level_callback(k, models[k], y, 0, [noise])
    C0 = _spd(rng, d, 2e-4 / d)
    if case == "da_pcn":
        e.set_proposal(1, None, scaling=0.01)
        prop = dict(kind="pcn", scaling=0.01)
    elif case == "da_grw_random":
        e.set_proposal(0, C0, scaling=1.0, adaptive=True, gamma=1.02, period=15)
        prop = dict(kind="grw", C=C0, scaling=1.0, adaptive=True, gamma=1.02, period=15)
    else:
        e.set_proposal(2, C0, t0=20, period=10, adaptive=True, gamma=1.02)
        prop = dict(kind="am", C0=C0, t0=20, period=10, adaptive=True, gamma=1.02)
    e.set_subchains(sl, randomize)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    e.close()
    us, ridx = _oracle_uniforms(seed, N, rows, sl, sl[0] if randomize else None)
    prior = orc.MVNPrior(pm, np.diag(pv))
    levels = [orc.CallableGaussianLevel(models[k], y, "iso", noise, prior) for k in range(nl)]
    res, _ = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, ridx)
    for k in range(nl):
        ref = res[k]
        sk = slice(1, None) if k == nl - 1 else slice(None)
        assert np.array_equal(outs[k][2], ref["accepted"][:, sk].T), "level %d accept masks differ" % k
        np.testing.assert_allclose(outs[k][1][:, :, 2], ref["logpost"][:, sk].T, rtol=1e-9 if case.startswith("mlda") else 1e-10)
        np.testing.assert_allclose(outs[k][0], np.swapaxes(ref["theta"][:, sk], 0, 1), rtol=1e-8, atol=1e-10)
    assert 0.0 < outs[0][2].mean() < 1.0


@pytest.mark.parametrize("case", ["dense_am", "dense_da_pcn", "joint_grw", "joint_mlda_am"])
def test_wide_dense_and_joint_priors_against_the_oracle(case):
    """65 .. 128 parameters under a Gaussian prior with a DENSE covariance (whitening product on the matrix cores, 8 x 8 fragment blocks)
    and under a JointPrior of normal and uniform components (support bounds tested per proposal): single level and hierarchies"""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, N = (100 if case.startswith("dense") else 90), 18
    rng = np.random.default_rng(31)
    truth = 0.3 * rng.standard_normal(d)
    hier = "da" in case or "mlda" in case
    ms = (30, 80) if "da" in case else ((24, 48, 96) if "mlda" in case else (120,))
    sl = [3] if "da" in case else ([3, 2] if "mlda" in case else [])
    nl = len(ms)
    As = [rng.standard_normal((m, d)) / np.sqrt(d) for m in ms]
    ys = [A @ truth + 0.1 * rng.standard_normal(len(A)) for A in As]
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    seed = 77
    e = Engine(N, d, seed=seed, n_levels=nl)
    if case.startswith("dense"):
        pm = 0.05 * rng.standard_normal(d)
        pc = _spd(rng, d, 1.0 / d) + 0.5 * np.eye(d)
        e.set_prior(pm, pc)
        prior = orc.MVNPrior(pm, pc)
    else:
        kinds = np.zeros(d, dtype=np.int32)
        kinds[[0, 2, 70, 89]] = 1  # uniform components, two of them beyond lane 64
        loc = np.where(kinds == 1, truth - 0.4, 0.0)
        scale = np.where(kinds == 1, 0.8, 1.0)
        e.set_prior_joint(kinds, loc, scale)
        prior = orc.JointPriorOracle(kinds, loc, scale)
    for k in range(nl):
        e.set_level(k, As[k], ys[k], 0, 0.01)
    C0 = _spd(rng, d, 4e-3 / d)
    if case.endswith("am"):
        e.set_proposal(2, C0, t0=16, period=8, adaptive=True, gamma=1.02)
        prop = dict(kind="am", C0=C0, t0=16, period=8, adaptive=True, gamma=1.02)
    elif case.endswith("pcn"):
        e.set_proposal(1, None, scaling=0.03)
        prop = dict(kind="pcn", scaling=0.03)
    else:
        e.set_proposal(0, C0, scaling=1.0, adaptive=True, gamma=1.02, period=8)
        prop = dict(kind="grw", C=C0, scaling=1.0, adaptive=True, gamma=1.02, period=8)
    if hier:
        e.set_subchains(sl, False)
        e.init(theta0)
        n_fine = 12
        rows = e.rows_per_level(n_fine)
        z, _ = e.set_export(rows[0])
        outs = e.run_levels_host(n_fine)
        e.close()
        us, _ = _oracle_uniforms(seed, N, rows, sl)
        levels = [orc.LinearGaussianLevel(As[k], ys[k], "iso", 0.01, prior) for k in range(nl)]
        res, _ = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, None)
        for k in range(nl):
            sk = slice(1, None) if k == nl - 1 else slice(None)
            assert np.array_equal(outs[k][2], res[k]["accepted"][:, sk].T), "level %d accept masks differ" % k
            np.testing.assert_allclose(outs[k][1][:, :, 2], res[k]["logpost"][:, sk].T, rtol=1e-9 if case.endswith("am") else 1e-10)
        assert 0.0 < outs[0][2].mean() < 1.0
    else:
        T = 70
        e.init(theta0)
        z, u = e.set_export(T)
        params, stats, acc = e.run_host(T)
        e.close()
        lvl = orc.LinearGaussianLevel(As[0], ys[0], "iso", 0.01, prior)
        ref = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
        assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
        np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-9 if case.endswith("am") else 1e-10)
        assert 0.0 < acc.mean() < 1.0


@pytest.mark.parametrize("hier", [False, True])
def test_wide_results_do_not_depend_on_sharding(hier):
    """chains keyed by global id at 100 parameters too: one engine with 35 chains == two engines with 16 and 19 (offsets 0 and 16), bit
    for bit -- what `bench.py --gpus N` and sample(distributed=True) rely on (single level: AdaptiveMetropolis across swaps; hier:
    three-level MLDA)"""
    from tinyda_amd.engine import Engine

    d, n_all, cut = 100, 35, 16
    rng = np.random.default_rng(12)
    truth = 0.3 * rng.standard_normal(d)
    ms = (24, 60, 150) if hier else (150,)
    As = [rng.standard_normal((m, d)) / np.sqrt(d) for m in ms]
    ys = [A @ truth + 0.1 * rng.standard_normal(len(A)) for A in As]
    theta0 = truth + 0.02 * rng.standard_normal((n_all, d))
    C0 = _spd(rng, d, 4e-3 / d)

    def run(n, off, th0):
        e = Engine(n, d, seed=9, chain_offset=off, n_levels=len(ms))
        e.set_prior(np.zeros(d), np.eye(d))
        for k in range(len(ms)):
            e.set_level(k, As[k], ys[k], 0, 0.01)
        e.set_proposal(2, C0, t0=20, period=20, adaptive=True)
        if hier:
            e.set_subchains([3, 2])
            e.init(th0)
            outs = e.run_levels_host(10)
            e.close()
            return [a for o in outs for a in o]
        e.init(th0)
        out = e.run_host(90)
        e.close()
        return list(out)

    full = run(n_all, 0, theta0)
    lo, hi = run(cut, 0, theta0[:cut]), run(n_all - cut, cut, theta0[cut:])
    for k in range(len(full)):
        assert np.array_equal(full[k][:, :cut], lo[k]) and np.array_equal(full[k][:, cut:], hi[k]), k


WIDE_SRC = """
__device__ double tda_forward(const double* theta, int dim, int o) {
  double s = 0.0;
  for (int j = 0; j < dim; ++j) s += (0.02 + 0.003 * ((o * 7 + j * 3) %% 11) + %(shift).4f) * theta[j];
  const double q = theta[o %% dim] * theta[(o + 70) %% dim];  // (a coupling that reaches parameters beyond lane 64)
  return sin(s) + %(coup).4f * q;
}
"""


def _wide_src_twin(shift, coup, m=23):
    def fn(theta):
        theta = np.atleast_2d(theta)
        N, d = theta.shape
        out = np.empty((N, m))
        for o in range(m):
            s = np.zeros(N)
            for j in range(d):  # same summation order as the device code
                s = s + (0.02 + 0.003 * ((o * 7 + j * 3) % 11) + float("%.4f" % shift)) * theta[:, j]
            out[:, o] = np.sin(s) + float("%.4f" % coup) * (theta[:, o % d] * theta[:, (o + 70) % d])
        return out
    return fn


@pytest.mark.parametrize("case", ["am", "pcn", "da_pcn", "mlda_am"])
def test_wide_source_defined_models_match_the_oracle(case):
    """source-defined (hiprtc-compiled) forward models at 96 parameters: the fused step kernel and the fused level action of the
    compiled module hold two parameters per lane; single level and hierarchies against the oracle running the NumPy twins"""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, m, N = 96, 23, 17
    rng = np.random.default_rng(21)
    truth = 0.3 * rng.standard_normal(d)
    hier = case.startswith("da") or case.startswith("mlda")
    cfgs = [dict(shift=0.0, coup=0.5)] if not hier else ([dict(shift=0.002, coup=0.4), dict(shift=0.0, coup=0.5)] if case.startswith("da")
                                                        else [dict(shift=0.004, coup=0.3), dict(shift=0.002, coup=0.4), dict(shift=0.0, coup=0.5)])
    nl = len(cfgs)
    twins = [_wide_src_twin(c["shift"], c["coup"]) for c in cfgs]
    y = twins[-1](truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    seed = 993
    e = Engine(N, d, seed=seed, n_levels=nl, block_steps=0 if hier else 33)
    e.set_prior(pm, np.diag(pv))
    for k in range(nl):
        e.set_level_source(k, WIDE_SRC % cfgs[k], y, 0, [0.05 ** 2])
    C0 = _spd(rng, d, 2e-4 / d)
    if case.endswith("am"):
        e.set_proposal(2, C0, t0=20, period=10, adaptive=True, gamma=1.02)
        prop = dict(kind="am", C0=C0, t0=20, period=10, adaptive=True, gamma=1.02)
    else:
        e.set_proposal(1, None, scaling=0.01, adaptive=True, gamma=1.02, period=15)
        prop = dict(kind="pcn", scaling=0.01, adaptive=True, gamma=1.02, period=15)
    prior = orc.MVNPrior(pm, np.diag(pv))
    levels = [orc.CallableGaussianLevel(twins[k], y, "iso", 0.05 ** 2, prior) for k in range(nl)]
    if hier:
        sl = [3] if nl == 2 else [3, 2]
        e.set_subchains(sl, False)
        e.init(theta0)
        n_fine = 10
        rows = e.rows_per_level(n_fine)
        z, _ = e.set_export(rows[0])
        outs = e.run_levels_host(n_fine)
        e.close()
        us, _ = _oracle_uniforms(seed, N, rows, sl)
        res, _ = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, None)
        for k in range(nl):
            sk = slice(1, None) if k == nl - 1 else slice(None)
            assert np.array_equal(outs[k][2], res[k]["accepted"][:, sk].T), "level %d accept masks differ" % k
            np.testing.assert_allclose(outs[k][1][:, :, 2], res[k]["logpost"][:, sk].T, rtol=1e-9 if case.endswith("am") else 1e-10)
        assert 0.0 < outs[0][2].mean() < 1.0
    else:
        T = 80
        e.init(theta0)
        z, u = e.set_export(T)
        params, stats, acc = e.run_host(T)
        e.close()
        ref = orc.run_mh(levels[0], prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
        assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
        np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-9 if case == "am" else 1e-10)
        assert 0.0 < acc.mean() < 1.0


@pytest.mark.parametrize("case", ["da_source", "mlda_mixed", "da_dep_pcn"])
def test_wide_hierarchy_with_error_model_over_external_models(case):
    """the dense error model at 96 parameters over source-defined levels, and over a linear surrogate + a batched host callback + a
    source-defined finest level (k_ext_aem_action / k_ext_aem_accept with the error-model row stride at least 128: a thread is an
    output AND a parameter) -- against the oracle's restatement of the reference's error-model chains running the NumPy twins"""
    from tests.test_gpu_multilevel import _oracle_uniforms
    from tinyda_amd.engine import Engine

    d, m, N = 96, 23, 15
    rng = np.random.default_rng(23)
    truth = 0.3 * rng.standard_normal(d)
    if case == "mlda_mixed":
        cfgs, sl, n_fine = [None, dict(shift=0.002, coup=0.4), dict(shift=0.0, coup=0.5)], [3, 2], 10
    else:
        cfgs, sl, n_fine = [dict(shift=0.002, coup=0.4), dict(shift=0.0, coup=0.5)], [1 if case == "da_dep_pcn" else 3], 24 if case == "da_dep_pcn" else 14
    aem = "state-dependent" if case == "da_dep_pcn" else "state-independent"
    nl = len(cfgs)
    Alin = np.array([[(0.02 + 0.003 * ((o * 7 + j * 3) % 11)) for j in range(d)] for o in range(m)])
    blin = 0.01 * np.arange(m)
    twins = [(lambda th: np.atleast_2d(th) @ Alin.T + blin) if c is None else _wide_src_twin(c["shift"], c["coup"]) for c in cfgs]
    y = twins[-1](truth)[0] + 0.05 * rng.standard_normal(m)
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    pm, pv = np.zeros(d), np.ones(d)
    var = 0.05 ** 2
    cov = var * np.eye(m)
    seed = 314
    e = Engine(N, d, seed=seed, n_levels=nl)
    e.set_prior(pm, np.diag(pv))
    for i, c in enumerate(cfgs):
        last = i == nl - 1
        if c is None:
            e.set_level(0, Alin, y, 3, cov, b=blin)
        elif case == "mlda_mixed" and i == 1:
            e.set_level_callback(1, twins[1], y, 3, cov)
        else:
            e.set_level_source(i, WIDE_SRC % c, y, 0 if last else 3, [var] if last else cov)
    C0 = _spd(rng, d, 2e-4 / d)
    if case == "da_dep_pcn":
        e.set_proposal(1, None, scaling=0.01)
        prop = dict(kind="pcn", scaling=0.01)
    else:
        e.set_proposal(0, C0, scaling=1.0)
        prop = dict(kind="grw", C=C0, scaling=1.0)
    e.set_subchains(sl, False)
    e.set_error_model(aem)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
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
        if i == nl - 1:
            np.testing.assert_allclose(outs[i][1][:, :, 2], (np.asarray(ref["logprior"]) + np.asarray(ref["loglike"]))[:, sk].T, rtol=1e-10)
    assert 0.0 < outs[nl - 1][2].mean() <= 1.0 and np.all(np.isfinite(bias)) and np.all(np.isfinite(P))
