"""GPU parity of DREAM(Z): replay of tinyDA's DREAMZ traces, Philox forward mode vs the oracle, shared archive."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc
from tests.test_oracle_dreamz import dreamz_inputs

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()
    return engine


def _setup(eng_mod, g, cfg, T, block=0, seed=3):
    N, d = g["theta0"].shape
    e = eng_mod.Engine(N, d, seed=seed, block_steps=block)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    if str(g["problem"]) == "linear":
        if "noise_cov" in g.files:  # dense observation covariance under DREAM(Z) (round 4)
            e.set_level(0, g["A"], g["data"], 2, g["noise_cov"])
        else:
            e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
    else:
        e.set_level_rosenbrock(0, float(g["rosen_a"]), float(g["rosen_b"]), 0.0, 1.0)
    e.set_proposal_dreamz(cfg["M0"], delta=cfg["delta"], b=cfg["b"], b_star=cfg["b_star"], nCR=cfg["nCR"],
                          adaptive=cfg["adaptive"], gamma=cfg["gamma"], period=cfg["period"], capacity=cfg["M0"] + T)
    return e


@pytest.mark.parametrize("name,block", [("g6_dreamz_linear", 0), ("g6_dreamz_linear", 7), ("g6_dreamz_rosen_adaptive", 0),
                                        ("g6_dreamz_empty_subspace", 0), ("g6_dreamz_linear_dense", 0), ("g6_dreamz_linear_dense", 5)])
def test_dreamz_replay(eng_mod, golden, name, block):
    g = golden(name)
    _, cfg, var = dreamz_inputs(g)
    T = g["u"].shape[1]
    e = _setup(eng_mod, g, cfg, T, block)
    e.set_archive(g["Z0"])
    e.init(g["theta0"])
    sw = lambda a: np.swapaxes(a, 0, 1)
    e.set_replay_dreamz(sw(var["r"]), sw(var["mcr"]), sw(var["sub_u"]), sw(var["forced"]), sw(var["e_u"]), sw(var["eps_n"]),
                        sw(var["u"]))
    _, st0 = e.current()
    params, stats, acc = e.run_host(T)
    assert np.array_equal(acc, g["accepted"][:, 1:].T), "%d accept flips" % int((acc != g["accepted"][:, 1:].T).sum())
    ref_post = g["logprior"] + g["loglike"]
    np.testing.assert_allclose(st0[:, 2], ref_post[:, 0], rtol=RTOL)
    np.testing.assert_allclose(stats[:, :, 2], ref_post[:, 1:].T, rtol=RTOL)
    np.testing.assert_allclose(params, sw(g["theta"][:, 1:]), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(e.proposal_state_scaling(), g["scaling"], rtol=1e-12)
    st = e.dreamz_state()
    np.testing.assert_allclose(st["pCR"], g["pCR"], rtol=1e-8)
    assert st["archive_rows"] == cfg["M0"] + T
    e.close()


def _philox_dreamz_variates(seed, N, T, d, delta, nCR, M0, pCR_fn, grow=True, chain_offset=0):
    """The oracle's restatement of the DREAM(Z) RNG contract (tda_kernels_dreamz.h, k_dreamz_draw): streams 4 (per chain and step:
    row pairs, crossover index), 5 (per parameter and step: crossover and e uniforms, 53 bits each)."""
    ps = orc.PhiloxStream(seed)
    chains = (np.arange(N) + chain_offset).astype(np.uint32)
    r = np.empty((N, T, delta, 2))
    u_mcr = np.empty((N, T))
    forced = np.empty((N, T))
    sub_u = np.empty((N, T, d))
    e_u = np.empty((N, T, d))
    u = np.empty((N, T))
    for t in range(T):
        M = np.uint64(M0 + (t if grow else 0))
        for i in range(delta):
            x0, x1, _, _ = ps.words(chains, np.uint32(t), np.uint32(4), np.uint32(i))
            r1 = (x0.astype(np.uint64) * M) >> np.uint64(32)
            r2 = (x1.astype(np.uint64) * (M - np.uint64(1))) >> np.uint64(32)
            r2 = r2 + (r2 >= r1)
            r[:, t, i, 0], r[:, t, i, 1] = r1, r2
        x0, x1, x2, _ = ps.words(chains, np.uint32(t), np.uint32(4), np.uint32(delta))
        u_mcr[:, t] = orc.u53(x0, x1)
        forced[:, t] = (x2.astype(np.uint64) * np.uint64(d)) >> np.uint64(32)
        for j in range(d):
            x0, x1, x2, x3 = ps.words(chains, np.uint32(t), np.uint32(5), np.uint32(delta + 1 + j))
            sub_u[:, t, j], e_u[:, t, j] = orc.u53(x0, x1), orc.u53(x2, x3)
        u[:, t] = ps.uniform(chains, t)
    return dict(r=r, u_mcr=u_mcr, forced=forced, sub_u=sub_u, e_u=e_u, u=u)


@pytest.mark.parametrize("d,delta", [(32, 1), (5, 2), (12, 1), (40, 3), (64, 1)])
def test_dreamz_philox_forward_rosenbrock32(eng_mod, d, delta):
    """BASELINE config-4 shape (d = 32 Rosenbrock chain, per-chain archive, non-adaptive so pCR stays uniform):
    the engine's own stream; integers / uniforms regenerated bit-exactly by the oracle, normals exported.  The other
    dimensions put every lane layout of k_dreamz_draw / k_dreamz_steps_wave (8, 16, 32, 64 padded parameters; 8 or 4 chains
    per wave; 1 to 4 parameters per lane) and several row-pair counts through the same comparison."""
    N, T, M0, nCR, seed = 40, 120, 64, 3, 2468
    prior_mean, prior_cov = np.zeros(d), np.eye(d)
    e = eng_mod.Engine(N, d, seed=seed)
    e.set_prior(prior_mean, prior_cov)
    e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
    e.set_proposal_dreamz(M0, delta=delta, nCR=nCR, capacity=M0 + T)
    rng = np.random.default_rng(1)
    Z0 = rng.standard_normal((N, M0, d))
    theta0 = 0.3 * rng.standard_normal((N, d))
    e.set_archive(Z0)
    e.init(theta0)
    eps, _ = e.set_export(T)
    params, stats, acc = e.run_host(T)
    v = _philox_dreamz_variates(seed, N, T, d, delta, nCR, M0, None)
    cdf = np.cumsum(np.full(nCR, 1.0 / nCR))
    mcr = np.minimum((v["u_mcr"][..., None] >= cdf).sum(-1), nCR - 1)
    var = dict(r=v["r"], mcr=mcr, sub_u=v["sub_u"], forced=v["forced"], e_u=v["e_u"], eps_n=np.swapaxes(eps, 0, 1), u=v["u"])
    level = orc.RosenbrockLevel(orc.MVNPrior(prior_mean, prior_cov))
    cfg = dict(M0=M0, delta=delta, nCR=nCR, adaptive=False, period=100, gamma=1.01, b=5e-2, b_star=1e-6)
    res = orc.run_dreamz(level, cfg, theta0, Z0, var)
    assert np.array_equal(acc, res["accepted"][:, 1:].T)
    np.testing.assert_allclose(stats[:, :, 2], res["logpost"][:, 1:].T, rtol=RTOL)
    assert 0.02 < acc.mean() < 0.9
    e.close()


@pytest.mark.parametrize("model,N,d", [("linear", 32, 8), ("rosenbrock", 32, 8), ("rosenbrock", 24, 12), ("linear", 40, 8)])
def test_dream_shared_archive_sharding_invariance(eng_mod, model, N, d):
    """DREAM (one archive for all chains, synchronised every K steps): two engines holding half of the chains each,
    exchanging rows through archive_take / archive_append in canonical (step, global chain) order, reproduce the
    single-engine run bit for bit.  This is the exchange the RCCL all-gather performs across GPUs.  Linear model (tile kernel)
    and the Rosenbrock chain (k_dreamz_steps_wave); chain counts that are not multiples of 16 exercise the padded chains and
    the strided archive copies."""
    T, M0, K, seed = 60, 24, 5, 99
    rng = np.random.default_rng(2)
    A = rng.standard_normal((12, d)) / np.sqrt(d)
    y = rng.standard_normal(12)
    Z0 = rng.standard_normal((M0, d))
    theta0 = 0.3 * rng.standard_normal((N, d))

    def make(n, off, th):
        e = eng_mod.Engine(n, d, seed=seed, chain_offset=off)
        e.set_prior(np.zeros(d), np.eye(d))
        if model == "linear":
            e.set_level(0, A, y, 0, 0.25)
        else:
            e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
        e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=True, period=20, gamma=1.02, shared=True, sync_every=K,
                              capacity=M0 + T * N)
        e.set_archive(Z0)
        e.init(th)
        return e

    one = make(N, 0, theta0)
    full = one.run_host(T)
    ref_state = one.dreamz_state()
    one.close()
    h = N // 2
    e0, e1 = make(h, 0, theta0[:h]), make(h, h, theta0[h:])
    for e in (e0, e1):
        e.set_archive_auto_append(False)
    outs0, outs1 = [], []
    for _ in range(T // K):
        outs0.append(e0.run_host(K))
        outs1.append(e1.run_host(K))
        r0, r1 = np.empty((K, h, d)), np.empty((K, h, d))
        assert e0.archive_take(r0) == K and e1.archive_take(r1) == K
        rows = np.concatenate([r0, r1], axis=1).reshape(K * N, d)  # all_gather along the chain axis
        e0.archive_append(rows)
        e1.archive_append(rows)
    for k in range(3):
        joined = np.concatenate([np.concatenate([a[k] for a in outs0]), np.concatenate([a[k] for a in outs1])], axis=1)
        assert np.array_equal(joined, full[k]), "shared-archive run depends on the sharding (record %d)" % k
    assert e0.dreamz_state()["archive_rows"] == ref_state["archive_rows"] == M0 + T * N
    np.testing.assert_array_equal(np.concatenate([e0.dreamz_state()["pCR"], e1.dreamz_state()["pCR"]]), ref_state["pCR"])
    e0.close()
    e1.close()


@pytest.mark.parametrize("model_kind", ["source", "callback"])
def test_dreamz_over_external_model_forward(eng_mod, model_kind):
    """DREAM(Z) over a non-linear model that lives outside the engine's kernels (source-defined / batched host callback):
    per step jump -> model -> accept -> archive append, against the oracle on the engine's own Philox stream."""
    from tests.test_gpu_usermodel import SRC, np_model

    d, m, N, T, M0, delta, nCR, seed = 5, 23, 19, 90, 24, 2, 3, 1357
    rng = np.random.default_rng(6)
    truth = 0.5 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(m)
    pm, pv = np.zeros(d), np.ones(d)
    e = eng_mod.Engine(N, d, seed=seed, block_steps=16)
    e.set_prior(pm, np.diag(pv))
    if model_kind == "source":
        e.set_level_source(0, SRC, y, 0, [0.05 ** 2])
    else:
        e.set_level_callback(0, np_model, y, 0, [0.05 ** 2])
    e.set_proposal_dreamz(M0, delta=delta, nCR=nCR, capacity=M0 + T)
    Z0 = truth + 0.3 * rng.standard_normal((N, M0, d))
    theta0 = truth + 0.1 * rng.standard_normal((N, d))
    e.set_archive(Z0)
    e.init(theta0)
    eps, _ = e.set_export(T)
    params, stats, acc = e.run_host(T)
    v = _philox_dreamz_variates(seed, N, T, d, delta, nCR, M0, None)
    cdf = np.cumsum(np.full(nCR, 1.0 / nCR))
    mcr = np.minimum((v["u_mcr"][..., None] >= cdf).sum(-1), nCR - 1)
    var = dict(r=v["r"], mcr=mcr, sub_u=v["sub_u"], forced=v["forced"], e_u=v["e_u"], eps_n=np.swapaxes(eps, 0, 1), u=v["u"])
    level = orc.CallableGaussianLevel(np_model, y, "iso", 0.05 ** 2, orc.MVNPrior(pm, np.diag(pv)))
    cfg = dict(M0=M0, delta=delta, nCR=nCR, adaptive=False, period=100, gamma=1.01, b=5e-2, b_star=1e-6)
    res = orc.run_dreamz(level, cfg, theta0, Z0, var)
    assert np.array_equal(acc, res["accepted"][:, 1:].T)
    np.testing.assert_allclose(stats[:, :, 2], res["logpost"][:, 1:].T, rtol=RTOL)
    assert 0.02 < acc.mean() < 0.95
    e.close()


def test_sample_api_dreamz_over_external_models():
    import scipy.stats as st

    import tinyda_amd as tda
    from tests.test_gpu_usermodel import SRC, np_model

    d, m = 5, 23
    rng = np.random.default_rng(9)
    truth = 0.5 * rng.standard_normal(d)
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(m)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    like = tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(m))
    for model, prop in ((tda.DeviceModel(SRC, m, reference=lambda th: np_model(th)[0]), tda.DREAMZ(30, delta=1, adaptive=True, period=25)),
                        (tda.BatchedModel(np_model, m), tda.DREAM(30, delta=2))):
        post = tda.Posterior(prior, like, model)
        res = tda.sample(post, prop, 120, n_chains=12, seed=3)
        assert res["sampler"] == "MH" and res.get("backend", "hip") != "host"
        link = res["chain_7"][-1]
        assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)
        assert np.mean(res["chain_7"].accepted[1:]) > 0.0


def test_sample_api_dreamz_with_uniform_prior_components():
    """DREAM(Z) under a JointPrior with uniform components over a callback model: states and archive stay inside the
    supports, the chains move, records carry the posterior."""
    import scipy.stats as st

    import tinyda_amd as tda
    from tests.test_gpu_usermodel import np_model

    d, m = 5, 23
    rng = np.random.default_rng(13)
    truth = np.array([0.3, -0.2, 0.25, 0.1, -0.1])
    y = np_model(truth)[0] + 0.05 * rng.standard_normal(m)
    prior = tda.JointPrior([st.uniform(-0.5, 1.0), st.norm(0, 1), st.uniform(-0.5, 1.0), st.norm(0, 0.5), st.uniform(-1, 2)])
    post = tda.Posterior(prior, tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(m)), tda.BatchedModel(np_model, m))
    np.random.seed(5)
    res = tda.sample(post, tda.DREAMZ(40, delta=1, adaptive=True, period=25), 150, n_chains=12, seed=3)
    assert res["sampler"] == "MH" and res.get("backend", "hip") != "host"
    th = np.stack([np.asarray(res["chain_%d" % i].parameters) for i in range(12)])
    assert (th[:, :, 0] >= -0.5).all() and (th[:, :, 0] <= 0.5).all() and (th[:, :, 2] >= -0.5).all() and (th[:, :, 2] <= 0.5).all()
    link = res["chain_7"][-1]
    assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)
    assert np.mean([np.mean(res["chain_%d" % i].accepted[1:]) for i in range(12)]) > 0.01


@pytest.mark.parametrize("name,block", [("g15_da_dreamz", 0), ("g15_da_dreamz", 5), ("g15_mlda_dreamz", 0), ("g15_mlda_dreamz", 7),
                                        ("g15_da_dreamz_random", 0), ("g15_da_dreamz_random", 6),
                                        ("g15_da_dreamz_aem", 0), ("g15_da_dreamz_aem", 5), ("g15_da_dreamz_aem_dep", 0), ("g15_mlda_dreamz_aem", 0),
                                        ("g15_mlda_dreamz_aem", 7), ("g15_da_dreamz_aem_m160", 0)])
def test_dreamz_below_a_hierarchy_replay(eng_mod, golden, name, block):
    """DREAMZ as the base proposal of Delayed Acceptance / MLDA (the reference's MLDA notebook configuration,
    examples/Multilevel Delayed Acceptance.ipynb cells 20-23; proposal.py:1583-1613, chain.py:404-444) on the device: traces
    recorded from tinyDA's DAChain / MLDAChain with DREAMZ, every draw of DREAMZ.make_proposal and every level's uniforms
    replayed -- accept flags of every level equal, log-posteriors to 1e-10, adapted scaling and crossover probabilities."""
    g = golden(name)
    nl = int(g["n_levels"])
    N, d = g["theta0"].shape
    sl = [int(v) for v in g["subchain_lengths"]]
    n_fine = g["th%d" % (nl - 1)].shape[1] - 1
    T0 = g["u0"].shape[1]
    e = eng_mod.Engine(N, d, seed=11, n_levels=nl, block_steps=block)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    aem = str(g["aem"]) if "aem" in g.files else None  # round 3: DREAMZ below DA / MLDA with the (dense) adaptive error model
    for k in range(nl):
        if aem is not None and k < nl - 1:  # AdaptiveGaussianLogLike on the coarse levels (chain.py:268-305)
            e.set_level(k, g["A%d" % k], g["y%d" % k], 3, float(g["noise_var"]) * np.eye(len(g["y%d" % k])), b=g["b%d" % k])
        else:
            e.set_level(k, g["A%d" % k], g["y%d" % k], 0, float(g["noise_var"]), b=g["b%d" % k] if aem is not None else None)
    e.set_proposal_dreamz(int(g["M0"]), delta=int(g["delta"]), b=float(g["b"]), b_star=float(g["b_star"]), nCR=int(g["nCR"]),
                          adaptive=bool(g["adaptive"]), gamma=float(g["gamma"]), period=int(g["period"]), capacity=int(g["M0"]) + T0)
    randomize = bool(g["randomize"])  # round 3: DREAMZ below Delayed Acceptance with randomize_subchain_length (chain.py:525-527)
    e.set_subchains(sl, randomize)
    if aem is not None:
        e.set_error_model(aem)
    e.set_archive(g["Z0"])
    e.init(g["theta0"])
    sw = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))
    e.set_replay_dreamz(sw(g["r"]), sw(g["mcr"]), sw(g["sub_u"]), sw(g["forced"]), sw(g["e_u"]), sw(g["eps_n"]), sw(g["u0"]))
    for k in range(1, nl):
        e.set_replay_level(k, sw(g["u%d" % k]))
    if randomize:
        e.set_replay_level(-1, sw(g["ridx"]))
    outs = e.run_levels_host(n_fine)
    for k in range(nl):
        p, s_, a = outs[k]
        off = 1 if k == nl - 1 else 0  # the finest trace carries the initial link
        ref_acc = g["acc%d" % k][:, off:].T
        assert np.array_equal(a, ref_acc), "level %d: %d accept flips" % (k, int((a != ref_acc).sum()))
        np.testing.assert_allclose(s_[:, :, 2], (g["lp%d" % k] + g["ll%d" % k])[:, off:].T, rtol=RTOL)
        np.testing.assert_allclose(p, sw(g["th%d" % k][:, off:]), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(e.proposal_state_scaling(), g["scaling"], rtol=1e-12)
    st = e.dreamz_state()
    np.testing.assert_allclose(st["pCR"], g["pCR"], rtol=1e-8)
    assert st["archive_rows"] == int(g["M0"]) + T0
    e.close()


def test_sample_dreamz_below_da_with_error_model_and_randomised_subchains():
    """sample([adaptive coarse, fine], DREAMZ(...), adaptive_error_model=..., randomize_subchain_length=...) stays on the device
    (round 3; the replays above pin the arithmetic on tinyDA's traces)"""
    import scipy.stats as stats

    import tinyda_amd as tda

    rng = np.random.default_rng(4)
    d, m = 5, 12
    Af = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = rng.standard_normal(d)
    y = Af @ truth + 0.2 * rng.standard_normal(m)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    cov = 0.04 * np.eye(m)
    coarse = tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), tda.LinearModel(Af + 0.1 * rng.standard_normal((m, d)), 0.05 * rng.standard_normal(m)))
    fine = tda.Posterior(prior, tda.GaussianLogLike(y, cov), tda.LinearModel(Af))
    res = tda.sample([coarse, fine], tda.DREAMZ(30, adaptive=True, period=20), 60, n_chains=16, subchain_length=4, seed=3,
                     adaptive_error_model="state-independent", backend="hip")
    assert res["sampler"] == "DA" and res["backend"] == "hip" and len(res["chain_fine_0"]) == 61 and len(res["chain_coarse_15"]) == 240
    link = res["chain_fine_3"][-1]
    np.testing.assert_allclose(link.likelihood, fine.create_link(link.parameters).likelihood, rtol=1e-10)
    plain = tda.Posterior(prior, tda.GaussianLogLike(y, cov), coarse.model)
    res = tda.sample([plain, fine], tda.DREAMZ(30), 40, n_chains=16, subchain_length=5, randomize_subchain_length=True, seed=3, backend="hip")
    assert res["backend"] == "hip" and len(res["chain_fine_0"]) == 41
    acc = np.mean([res["chain_fine_%d" % i].accepted[1:].mean() for i in range(16)])
    assert 0.02 < acc < 0.98


def test_sample_dreamz_below_mlda_on_device():
    """tda.sample([3 posteriors], DREAMZ(Z_method='lhs', adaptive=True), subchain_length=...) lowers to the engine: linear
    levels, and the notebook's kind of model (a plain Python callable returning (output, qoi)) behind the callback path"""
    import scipy.stats as stats

    import tinyda_amd as tda

    d = 4
    rng = np.random.default_rng(8)
    truth = rng.standard_normal(d) * 0.3
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    As = [rng.standard_normal((m, d)) / 2 for m in (6, 10, 16)]
    ys = [A @ truth + 0.05 * rng.standard_normal(A.shape[0]) for A in As]
    lin = [tda.Posterior(prior, tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(len(y))), tda.LinearModel(A)) for A, y in zip(As, ys)]
    res = tda.sample(lin, tda.DREAMZ(M0=40, delta=1, Z_method="lhs", adaptive=True, period=10), iterations=30, n_chains=8,
                     initial_parameters=truth, subchain_length=3, seed=2, backend="hip")
    assert res["sampler"] == "MLDA" and res["backend"] == "hip" and len(res["chain_l2_0"]) == 31 and len(res["chain_l0_0"]) == 270
    fine = tda.get_samples(res, level=2)
    assert np.isfinite(fine["chain_0"]).all()
    # same seed -> same run (the LHS archive is drawn on the host from a generator keyed by (seed, global chain id))
    res2 = tda.sample(lin, tda.DREAMZ(M0=40, delta=1, Z_method="lhs", adaptive=True, period=10), iterations=30, n_chains=8,
                      initial_parameters=truth, subchain_length=3, seed=2, backend="hip")
    assert np.array_equal(tda.get_samples(res2, level=2)["chain_3"], fine["chain_3"])
    opaque = [tda.Posterior(prior, tda.GaussianLogLike(y, 0.05 ** 2 * np.eye(len(y))), lambda th, A=A: (np.tanh(A @ th), True))
              for A, y in zip(As, ys)]
    res3 = tda.sample(opaque, tda.DREAMZ(M0=40, Z_method="lhs", adaptive=True, period=10), iterations=12, n_chains=4,
                      initial_parameters=truth, subchain_length=2, seed=3)
    assert res3["backend"] == "hip" and len(res3["chain_l2_1"]) == 13
    assert res3["chain_l2_0"][5].qoi is True


def test_dream_overlapped_exchange_is_sharding_invariant(eng_mod):
    """run_shared_dream(overlap=True): the rows of block b reach the archive before block b + 2 (their exchange runs under
    block b + 1).  A single engine driven by the pipeline equals two half-size engines exchanging by hand with the same lag,
    bit for bit; the engine runs on a torch stream, nothing in the loop waits on the host."""
    import torch

    from tinyda_amd import distributed as tdist

    d, N, T, M0, K, seed = 8, 32, 63, 24, 5, 77  # 63 = 12 full blocks + a ragged one
    rng = np.random.default_rng(4)
    A = rng.standard_normal((12, d)) / np.sqrt(d)
    y = rng.standard_normal(12)
    Z0 = rng.standard_normal((M0, d))
    theta0 = 0.3 * rng.standard_normal((N, d))

    def make(n, off, th, stream=None):
        e = eng_mod.Engine(n, d, seed=seed, chain_offset=off, stream=stream)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level(0, A, y, 0, 0.25)
        e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=True, period=20, gamma=1.02, shared=True, sync_every=K, capacity=M0 + T * N)
        e.set_archive(Z0)
        e.init(th)
        return e

    ts = torch.cuda.Stream()
    one = make(N, 0, theta0, ts.cuda_stream)
    dev = torch.device("cuda", 0)
    p = torch.zeros((T, N, d), dtype=torch.float64, device=dev)
    s_ = torch.zeros((T, N, 3), dtype=torch.float64, device=dev)
    a = torch.zeros((T, N), dtype=torch.uint8, device=dev)
    tdist.run_shared_dream(one, T, K, p, s_, a, overlap=True, stream=ts)
    ts.synchronize()
    assert one.dreamz_state()["archive_rows"] == M0 + T * N
    ref_pcr = one.dreamz_state()["pCR"]
    one.close()

    h = N // 2
    e0, e1 = make(h, 0, theta0[:h]), make(h, h, theta0[h:])
    for e in (e0, e1):
        e.set_archive_auto_append(False)
    pend, got0, got1 = [], [], []
    done = 0
    while done < T:
        k = min(K, T - done)
        if len(pend) == 2:
            rows = pend.pop(0)
            e0.archive_append(rows)
            e1.archive_append(rows)
        got0.append(e0.run_host(k))
        got1.append(e1.run_host(k))
        r0, r1 = np.empty((k, h, d)), np.empty((k, h, d))
        assert e0.archive_take(r0) == k and e1.archive_take(r1) == k
        pend.append(np.concatenate([r0, r1], axis=1).reshape(k * N, d))
        done += k
    for rows in pend:
        e0.archive_append(rows)
        e1.archive_append(rows)
    for k, full in enumerate((p, s_, a)):
        joined = np.concatenate([np.concatenate([g[k] for g in got0]), np.concatenate([g[k] for g in got1])], axis=1)
        assert np.array_equal(joined, full.cpu().numpy()), "the overlapped exchange depends on the sharding (record %d)" % k
    np.testing.assert_array_equal(np.concatenate([e0.dreamz_state()["pCR"], e1.dreamz_state()["pCR"]]), ref_pcr)
    e0.close()
    e1.close()


@pytest.mark.parametrize("model,adaptive", [("linear", False), ("rosenbrock", False), ("rosenbrock", True), ("linear", True)])
def test_dream_distributed_archive_equals_the_replicated_one(eng_mod, model, adaptive):
    """tda_engine_set_archive_peers: every rank keeps only the rows of its own chains, proposals read the owners' segments in
    place.  Two engines of this process (half of the chains each, segments exchanged as device pointers) against one engine with
    the replicated archive: without adaptation bit for bit; with it the crossover sums are added per rank instead of per row, so
    accept flags must agree and densities / pCR to rounding."""
    d, N, T, M0, K, seed = 8, 32, 63, 24, 5, 123
    rng = np.random.default_rng(12)
    A = rng.standard_normal((12, d)) / np.sqrt(d)
    y = rng.standard_normal(12)
    Z0 = rng.standard_normal((M0, d))
    theta0 = 0.3 * rng.standard_normal((N, d))

    def make(n, off, th):
        e = eng_mod.Engine(n, d, seed=seed, chain_offset=off)
        e.set_prior(np.zeros(d), np.eye(d))
        if model == "linear":
            e.set_level(0, A, y, 0, 0.25)
        else:
            e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
        e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=adaptive, period=20, gamma=1.02, shared=True, sync_every=K, capacity=M0 + T * N)
        e.set_archive(Z0)
        e.init(th)
        return e

    one = make(N, 0, theta0)
    full = one.run_host(T)
    ref_pcr = one.dreamz_state()["pCR"]
    one.close()
    h = N // 2
    e0, e1 = make(h, 0, theta0[:h]), make(h, h, theta0[h:])
    ptrs = [e0.archive_pointer(), e1.archive_pointer()]
    e0.set_archive_peers(2, 0, pointers=ptrs)
    e1.set_archive_peers(2, 1, pointers=ptrs)
    with pytest.raises(Exception):
        e0.run_host(K + 1)  # more than one exchange interval per call
    with pytest.raises(Exception, match="distributed"):
        e0.archive_append(np.zeros((N, d)))  # the replicated-archive exchange does not apply
    with pytest.raises(Exception, match="distributed"):
        e0.archive_take(np.empty((1, h, d)))
    outs0, outs1 = [], []
    done = 0
    while done < T:
        k = min(K, T - done)
        outs0.append(e0.run_host(k))
        outs1.append(e1.run_host(k))
        sums = e0.archive_local_sums() + e1.archive_local_sums()
        e0.archive_publish(sums)
        e1.archive_publish(sums)
        done += k
    assert e0.dreamz_state()["archive_rows"] == M0 + T * N
    joined = [np.concatenate([np.concatenate([a[k] for a in outs0]), np.concatenate([a[k] for a in outs1])], axis=1) for k in range(3)]
    pcr = np.concatenate([e0.dreamz_state()["pCR"], e1.dreamz_state()["pCR"]])
    e0.close()
    e1.close()
    assert np.array_equal(joined[2], full[2]), "%d accept flips" % int((joined[2] != full[2]).sum())
    if adaptive:
        np.testing.assert_allclose(joined[0], full[0], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(joined[1], full[1], rtol=1e-9)
        np.testing.assert_allclose(pcr, ref_pcr, rtol=1e-9)
    else:
        assert np.array_equal(joined[0], full[0]) and np.array_equal(joined[1], full[1])


def test_dream_distributed_archive_checkpoint_resume(eng_mod):
    """get_state / set_state with the distributed archive: the blob carries the rank's segment and the position of the publish
    protocol, the peer mappings are set up again by the restoring process; resumed ranks continue bit for bit (adaptive run:
    the deferred column sums and the pending-adaptation bookkeeping are part of the state)."""
    d, N, M0, K, seed = 8, 32, 24, 5, 77
    rng = np.random.default_rng(4)
    Z0 = rng.standard_normal((M0, d))
    theta0 = 0.3 * rng.standard_normal((N, d))
    h = N // 2

    def make(off):
        e = eng_mod.Engine(h, d, seed=seed, chain_offset=off)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
        e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=True, period=20, gamma=1.02, shared=True, sync_every=K, capacity=M0 + 60 * N)
        e.set_archive(Z0)
        e.init(theta0[off:off + h])
        return e

    def pair():
        es = [make(0), make(h)]
        ptrs = [e.archive_pointer() for e in es]
        for r, e in enumerate(es):
            e.set_archive_peers(2, r, pointers=ptrs)
        return es

    def advance(es, steps):
        outs, done = [[], []], 0
        while done < steps:
            k = min(K, steps - done)
            for r, e in enumerate(es):
                outs[r].append(e.run_host(k))
            sums = es[0].archive_local_sums() + es[1].archive_local_sums()
            for e in es:
                e.archive_publish(sums)
            done += k
        return [np.concatenate([np.concatenate([o[i] for o in outs[r]]) for r in range(2)], axis=1) for i in range(3)]

    first = pair()
    advance(first, 35)  # past one adaptation boundary, not on a multiple of the period
    blobs = [e.get_state() for e in first]
    ref = advance(first, 25)
    ref_rows = first[0].dreamz_state()["archive_rows"]
    ref_pcr = [e.dreamz_state()["pCR"] for e in first]
    for e in first:
        e.close()
    second = pair()
    for e, b in zip(second, blobs):
        e.set_state(b)
        assert e.counters()[0] == 35  # the engine's own step counter (run_peer_dream finds its adaptation boundaries from it)
    got = advance(second, 25)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)
    assert second[0].dreamz_state()["archive_rows"] == ref_rows == M0 + 60 * N
    for e, pc in zip(second, ref_pcr):
        np.testing.assert_array_equal(e.dreamz_state()["pCR"], pc)
        e.close()


@pytest.mark.parametrize("model,adaptive", [("rosenbrock", False), ("linear", True)])
def test_dream_distributed_archive_lagged_publish(eng_mod, model, adaptive):
    """two unpublished blocks: block b runs while the collective of block b - 1 is in flight, its rows become visible from block
    b + 2 -- the lag of run_shared_dream(overlap=True).  Two engines with distributed segments against ONE engine with the
    replicated archive driven by that pipeline."""
    import torch

    from tinyda_amd import distributed as tdist

    d, N, T, M0, K, seed, period = 8, 32, 63, 24, 5, 77, 20
    rng = np.random.default_rng(14)
    A = rng.standard_normal((12, d)) / np.sqrt(d)
    y = rng.standard_normal(12)
    Z0 = rng.standard_normal((M0, d))
    theta0 = 0.3 * rng.standard_normal((N, d))

    def make(n, off, th, stream=None):
        e = eng_mod.Engine(n, d, seed=seed, chain_offset=off, stream=stream)
        e.set_prior(np.zeros(d), np.eye(d))
        if model == "linear":
            e.set_level(0, A, y, 0, 0.25)
        else:
            e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
        e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=adaptive, period=period, gamma=1.02, shared=True, sync_every=K, capacity=M0 + T * N)
        e.set_archive(Z0)
        e.init(th)
        return e

    ts = torch.cuda.Stream()
    one = make(N, 0, theta0, ts.cuda_stream)
    dev = torch.device("cuda", 0)
    p = torch.zeros((T, N, d), dtype=torch.float64, device=dev)
    s_ = torch.zeros((T, N, 3), dtype=torch.float64, device=dev)
    a = torch.zeros((T, N), dtype=torch.uint8, device=dev)
    tdist.run_shared_dream(one, T, K, p, s_, a, overlap=True, stream=ts)
    ts.synchronize()
    full = [p.cpu().numpy(), s_.cpu().numpy(), a.cpu().numpy()]
    ref_pcr = one.dreamz_state()["pCR"]
    one.close()

    h = N // 2
    e0, e1 = make(h, 0, theta0[:h]), make(h, h, theta0[h:])
    ptrs = [e0.archive_pointer(), e1.archive_pointer()]
    e0.set_archive_peers(2, 0, pointers=ptrs)
    e1.set_archive_peers(2, 1, pointers=ptrs)
    outs0, outs1, unpublished, done, t = [], [], 0, 0, 0
    while done < T:
        k = min(K, T - done, period - t % period)
        if unpublished == 2:
            e0.archive_publish(None)
            e1.archive_publish(None)
            unpublished -= 1
        outs0.append(e0.run_host(k))
        outs1.append(e1.run_host(k))
        unpublished += 1
        if adaptive and (t + k) % period == 0:
            sums = e0.archive_local_sums() + e1.archive_local_sums()
            e0.archive_publish(sums)
            e1.archive_publish(sums)
            unpublished -= 1
        if unpublished == 2:
            with pytest.raises(Exception):
                e0.run_host(1)  # a third unpublished block
        done += k
        t += k
    while unpublished:
        e0.archive_publish(None)
        e1.archive_publish(None)
        unpublished -= 1
    assert e0.dreamz_state()["archive_rows"] == M0 + T * N
    joined = [np.concatenate([np.concatenate([x[k] for x in outs0]), np.concatenate([x[k] for x in outs1])], axis=1) for k in range(3)]
    pcr = np.concatenate([e0.dreamz_state()["pCR"], e1.dreamz_state()["pCR"]])
    e0.close()
    e1.close()
    assert np.array_equal(joined[2], full[2]), "%d accept flips" % int((joined[2] != full[2]).sum())
    if adaptive:
        np.testing.assert_allclose(joined[1], full[1], rtol=1e-9)
        np.testing.assert_allclose(pcr, ref_pcr, rtol=1e-9)
    else:
        assert np.array_equal(joined[0], full[0]) and np.array_equal(joined[1], full[1])


def test_sample_api_with_the_distributed_archive():
    """tda.sample(..., DREAM(...), shared_archive='distributed'): one process is one rank -- the protocol of the N > 1 run with
    its own segment only; seeded runs repeat, the chains move, the lagged variant runs on its stream"""
    import scipy.stats as stats

    import tinyda_amd as tda

    d = 6
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    rng = np.random.default_rng(3)
    A = rng.standard_normal((10, d)) / 2
    post = tda.Posterior(prior, tda.GaussianLogLike(A @ (0.3 * rng.standard_normal(d)), 0.04 * np.eye(10)), tda.LinearModel(A))
    runs = []
    for overlap in (False, False, True):
        res = tda.sample(post, tda.DREAM(M0=30, delta=1, adaptive=True, period=32), iterations=70, n_chains=32, seed=5, backend="hip",
                         shared_archive="distributed", overlap_archive_exchange=overlap)
        x = tda.get_samples(res)
        runs.append(np.stack([x["chain_%d" % c] for c in range(32)]))
        assert res["backend"] == "hip" and np.isfinite(runs[-1]).all()
    assert np.array_equal(runs[0], runs[1])
    assert (np.abs(np.diff(runs[0], axis=1)).sum(axis=(1, 2)) > 0).all()  # every chain moved
    assert runs[2].shape == runs[0].shape
    with pytest.raises(ValueError):
        tda.sample(post, tda.DREAM(M0=30), iterations=2, n_chains=16, seed=5, shared_archive="scattered")
