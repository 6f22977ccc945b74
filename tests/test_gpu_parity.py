"""GPU parity: the HIP engine (through the C-ABI) against the oracle and the reference's golden vectors.

Bar: accept masks bit-exact, log-posterior within 1e-10 relative (BASELINE.json north_star).
"""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-10


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()  # fail loudly if the HIP library is not built
    return engine


def _mk_engine(eng_mod, g, n_chains, d, kind, **kw):
    e = eng_mod.Engine(n_chains, d, seed=kw.pop("seed", 1234), chain_offset=kw.pop("chain_offset", 0),
                       block_steps=kw.pop("block_steps", 0))
    e.set_prior(g["prior_mean"], g["prior_cov"])
    if "noise_var" in g:
        e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
    else:
        nk = str(g["noise_kind"])
        if nk == "iso":
            e.set_level(0, g["A"], g["data"], 0, float(g["noise_cov"][0]))
        elif nk == "diag":
            e.set_level(0, g["A"], g["data"], 1, g["noise_cov"])
        else:
            e.set_level(0, g["A"], g["data"], 2, g["noise_cov"])
    return e


def _compare_with_golden(params, stats, acc, th0_stats, g, tight=True):
    # engine records are [T, N, .]; golden traces are [N, T+1, .] with the initial link first
    acc_ref = np.swapaxes(g["accepted"][:, 1:], 0, 1)
    assert np.array_equal(acc, acc_ref), "accept masks differ: %d flips" % int((acc != acc_ref).sum())
    np.testing.assert_allclose(th0_stats[:, 2], g["logpost"][:, 0], rtol=RTOL)
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=RTOL)
    # the stated bar is on the log-posterior; its two summands and the states are held to the same 1e-10 wherever the proposal
    # covariance is well conditioned (`tight`: g1, g2_am_c2, g2b -- VERDICT r3 item 9) and a decade looser only for the small AM
    # fixtures, which swap in a nearly singular C (16 samples in 8 dims) whose factorisation amplifies last-bit differences
    # between LAPACK and the device's Cholesky.
    tol = 1e-10 if tight else 1e-9
    np.testing.assert_allclose(stats[:, :, 1], np.swapaxes(g["loglike"][:, 1:], 0, 1), rtol=tol)
    np.testing.assert_allclose(stats[:, :, 0], np.swapaxes(g["logprior"][:, 1:], 0, 1), rtol=tol, atol=1e-12)
    np.testing.assert_allclose(params, np.swapaxes(g["theta"][:, 1:], 0, 1), rtol=tol, atol=1e-11)  # (parameters near zero: an absolute floor)


def test_device_rng_matches_contract(eng_mod):
    """Philox words identical (uniforms bit-equal); Box-Muller normals to 1e-14."""
    N, d, seed, off = 48, 7, 0x1234567887654321, 1000
    e = eng_mod.Engine(N, d, seed=seed, chain_offset=off)
    ps = orc.PhiloxStream(seed)
    chains = np.arange(off, off + N)
    for step in (0, 1, 77, 2**31 + 5):
        z, u = e.rng_probe(step)
        assert np.array_equal(u, ps.uniform(chains, step)), "accept-uniform stream differs at step %d" % step
        np.testing.assert_allclose(z, ps.normals(chains, step, d), rtol=1e-13, atol=1e-14)
    e.close()


@pytest.mark.parametrize("d,m,noise,prior", [(2, 50, "iso", "identity"), (8, 16, "iso", "identity"),
                                              (6, 20, "diag", "general"), (17, 33, "diag", "diagonal"),
                                              (32, 100, "iso", "general"), (64, 1024, "iso", "identity"),
                                              (64, 300, "diag", "general")])
def test_evaluate_matches_oracle(eng_mod, d, m, noise, prior):
    rng = np.random.default_rng(d * 1000 + m)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    b = 0.1 * rng.standard_normal(m)
    y = rng.standard_normal(m)
    if prior == "identity":
        pm, pc = np.zeros(d), np.eye(d)
    elif prior == "diagonal":
        pm, pc = 0.2 * rng.standard_normal(d), np.diag(0.5 + rng.random(d))
    else:
        B = rng.standard_normal((d, d)) / np.sqrt(d)
        pm, pc = 0.2 * rng.standard_normal(d), B @ B.T + 0.5 * np.eye(d)
    nz = 0.3 if noise == "iso" else 0.1 + rng.random(m)
    N = 37
    e = eng_mod.Engine(N, d)
    e.set_prior(pm, pc)
    e.set_level(0, A, y, 0 if noise == "iso" else 1, nz, b=b)
    theta = rng.standard_normal((N, d))
    st = e.evaluate(theta)
    lvl = orc.LinearGaussianLevel(A, y, noise, nz, orc.MVNPrior(pm, pc), b=b)
    lp, ll, _ = lvl.evaluate(theta)
    np.testing.assert_allclose(st[:, 0], lp, rtol=1e-12)
    np.testing.assert_allclose(st[:, 1], ll, rtol=1e-12)
    np.testing.assert_allclose(st[:, 2], lp + ll, rtol=1e-12)
    e.close()


@pytest.mark.parametrize("d,m", [(6, 20), (8, 16), (17, 33), (32, 100), (64, 300), (64, 1024)])
def test_evaluate_dense_noise_matches_oracle(eng_mod, d, m):
    """DefaultGaussianLogLike (dense data covariance, distributions.py:246-301): r^T Sigma^-1 r on the MFMA path."""
    rng = np.random.default_rng(d * 7 + m)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    y = rng.standard_normal(m)
    Lc = 0.3 * np.eye(m) + 0.02 * np.tril(rng.standard_normal((m, m)))
    cov = Lc @ Lc.T
    N = 29
    e = eng_mod.Engine(N, d)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 2, cov)
    theta = rng.standard_normal((N, d))
    st = e.evaluate(theta)
    lvl = orc.LinearGaussianLevel(A, y, "dense", cov, orc.MVNPrior(np.zeros(d), np.eye(d)))
    lp, ll, _ = lvl.evaluate(theta)
    np.testing.assert_allclose(st[:, 1], ll, rtol=1e-11)
    np.testing.assert_allclose(st[:, 2], lp + ll, rtol=1e-11)
    e.close()


def test_golden_g1_grw_adaptive_replay(eng_mod, golden):
    g = golden("g1_basic_sampler")
    N, T1, d = g["theta"].shape
    e = _mk_engine(eng_mod, g, N, d, "grw")
    e.set_proposal(0, g["C"], scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]),
                   period=int(g["period"]))
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    _, st0 = e.current()
    params, stats, acc = e.run_host(T1 - 1)
    _compare_with_golden(params, stats, acc, st0, g)
    np.testing.assert_allclose(e.proposal_state()["scaling"], g["scaling_hist"][:, -1], rtol=1e-12)
    e.close()


@pytest.mark.parametrize("name,block", [("g2_am_small", 0), ("g2_am_small", 5), ("g2_am_small_adaptive", 0),
                                        ("g2_am_diag_genprior", 0), ("g2_am_dense", 0), ("g2_am_c2", 0),
                                        ("g2_am_d96", 0), ("g2_am_d96", 37)])  # (96 parameters: the 65 .. 128-parameter path, round 5)
def test_golden_g2_am_replay(eng_mod, golden, name, block):
    g = golden(name)
    N, T1, d = g["theta"].shape
    e = _mk_engine(eng_mod, g, N, d, "am", block_steps=block)
    e.set_proposal(2, g["C0"], adaptive=bool(g["adaptive"]), gamma=float(g["gamma"]), period=int(g["period"]),
                   sd=float(g["sd"]), epsilon=float(g["epsilon"]), t0=int(g["t0"]))
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    _, st0 = e.current()
    params, stats, acc = e.run_host(T1 - 1)
    # (g2_am_d96: masks exact and log-posterior to 1e-10 like every trace; its 96-parameter states after the two swaps to 1e-9 --
    # 3 of 57 600 entries read 9e-10 -- like the small fixtures: the factor of a 200-sample covariance in 96 dimensions is not LAPACK's bit for bit)
    _compare_with_golden(params, stats, acc, st0, g, tight=(name == "g2_am_c2"))
    ps = e.proposal_state(want_am=True)
    assert not e.flags().any()
    np.testing.assert_allclose(ps["am_mu"], g["mu_hist"][:, -1], rtol=1e-9)
    np.testing.assert_allclose(ps["am_sigma"], g["sigma_hist"][:, -1], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ps["C"], g["C_hist"][:, -1], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ps["scaling"], g["scaling_hist"][:, -1], rtol=1e-12)
    e.close()


@pytest.mark.parametrize("name", ["g2_am_small", "g2_am_small_adaptive", "g2_am_diag_genprior", "g2_am_c2"])
def test_golden_g2_am_replay_block_moments(eng_mod, golden, name):
    """AdaptiveMetropolis(block_moments=True): the covariance is updated once per block in closed form on the matrix
    cores.  Same accept masks as the reference's traces; the log-posterior agrees to 1e-8 instead of 1e-10, because
    the closed form does not reproduce the rounding error of the reference recursion (which cancels t mu mu^T against
    (t+1) mu' mu'^T); the mean is still bit-comparable."""
    g = golden(name)
    N, T1, d = g["theta"].shape
    e = _mk_engine(eng_mod, g, N, d, "am")
    e.set_proposal(2, g["C0"], adaptive=bool(g["adaptive"]), gamma=float(g["gamma"]), period=int(g["period"]),
                   sd=float(g["sd"]), epsilon=float(g["epsilon"]), t0=int(g["t0"]), block_moments=True)
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    params, stats, acc = e.run_host(T1 - 1)
    assert np.array_equal(acc, np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-8)
    ps = e.proposal_state(want_am=True)
    assert not e.flags().any()
    np.testing.assert_allclose(ps["am_mu"], g["mu_hist"][:, -1], rtol=1e-9, atol=1e-11)
    sig_ref = g["sigma_hist"][:, -1]
    np.testing.assert_allclose(ps["am_sigma"], sig_ref, rtol=1e-6, atol=1e-7 * np.abs(sig_ref).max())
    e.close()


def test_block_moments_forward_mode_vs_oracle(eng_mod):
    """block_moments at the C2a shape in Philox mode, ragged blocks (block_steps = 37 against period 50)."""
    d, m, N, T = 64, 256, 40, 230
    A, th, y = _c2_problem(d, m)
    theta0 = th + 0.02 * np.random.default_rng(3).standard_normal((N, d))
    e = eng_mod.Engine(N, d, seed=99, block_steps=37)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=50, period=50, block_moments=True)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    ps = e.proposal_state(want_am=True)
    e.close()
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    ref = orc.run_mh(lvl, dict(kind="am", C0=1e-4 * np.eye(d), t0=50, period=50), theta0,
                     np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-9)
    np.testing.assert_allclose(ps["am_sigma"], ref["am_sigma"], rtol=1e-8, atol=1e-8 * np.abs(ref["am_sigma"]).max())


def test_golden_g2b_pcn_replay(eng_mod, golden):
    g = golden("g2b_pcn")
    N, T1, d = g["theta"].shape
    e = _mk_engine(eng_mod, g, N, d, "pcn")
    e.set_proposal(1, None, scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]),
                   period=int(g["period"]))
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    _, st0 = e.current()
    params, stats, acc = e.run_host(T1 - 1)
    _compare_with_golden(params, stats, acc, st0, g)
    np.testing.assert_allclose(e.proposal_state()["scaling"], g["scaling_hist"][:, -1], rtol=1e-12)
    e.close()


def _c2_problem(d=64, m=1024, seed=1, sigma=0.1):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d)) / 8
    theta_true = rng.standard_normal(d)
    y = A @ theta_true + sigma * rng.standard_normal(m)
    return A, theta_true, y


def test_philox_forward_mode_vs_oracle(eng_mod):
    """Engine draws its own Philox variates and exports them; the oracle replays the identical stream."""
    d, m, N, T = 64, 1024, 48, 250
    A, theta_true, y = _c2_problem(d, m)
    rng = np.random.default_rng(5)
    theta0 = theta_true + 0.02 * rng.standard_normal((N, d))
    e = eng_mod.Engine(N, d, seed=99)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100, adaptive=True)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    prop = dict(kind="am", C0=1e-4 * np.eye(d), t0=100, period=100, adaptive=True)
    res = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(res["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(res["logpost"][:, 1:], 0, 1), rtol=RTOL)
    # the exported variates are the contract stream
    ps = orc.PhiloxStream(99)
    np.testing.assert_allclose(z[17], ps.normals(np.arange(N), 17, d), rtol=1e-13, atol=1e-14)
    assert np.array_equal(u[17], ps.uniform(np.arange(N), 17))
    e.close()


@pytest.mark.parametrize("d,m,kind,block", [(2, 20, "grw", 0), (7, 33, "am", 0), (17, 40, "am", 23), (32, 64, "pcn", 0),
                                             (33, 70, "am", 16)])
def test_philox_forward_mode_small_dims(eng_mod, d, m, kind, block):
    """The split proposal path (k_rng on the second stream, k_apply) at every padded dimension (8, 16, 32, 64) and with
    ragged blocks, against the oracle on the exported stream."""
    N, T = 21, 130
    A, theta_true, y = _c2_problem(d, m, seed=d)
    theta0 = theta_true + 0.05 * np.random.default_rng(d).standard_normal((N, d))
    e = eng_mod.Engine(N, d, seed=5 + d, chain_offset=3, block_steps=block)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    C0 = 1e-3 * (np.eye(d) + 0.3 * np.ones((d, d)) / d)
    if kind == "grw":
        e.set_proposal(0, C0, scaling=0.7, adaptive=True, period=20)
        prop = dict(kind="grw", C=C0, scaling=0.7, adaptive=True, period=20)
    elif kind == "pcn":
        e.set_proposal(1, None, scaling=0.05, adaptive=True, period=25)
        prop = dict(kind="pcn", scaling=0.05, adaptive=True, period=25)
    else:
        e.set_proposal(2, C0, t0=40, period=20)
        prop = dict(kind="am", C0=C0, t0=40, period=20)
    e.init(theta0)
    z, u = e.set_export(T)
    params, stats, acc = e.run_host(T)
    e.close()
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    res = orc.run_mh(lvl, prop, theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(res["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(res["logpost"][:, 1:], 0, 1), rtol=RTOL)
    ps = orc.PhiloxStream(5 + d)
    np.testing.assert_allclose(z[T - 1], ps.normals(np.arange(3, 3 + N), T - 1, d), rtol=1e-13, atol=1e-14)
    assert np.array_equal(u[T - 1], ps.uniform(np.arange(3, 3 + N), T - 1))


def test_results_do_not_depend_on_sharding(eng_mod):
    """Chains keyed by global id: one engine with 32 chains == two engines with 16 (offsets 0 and 16)."""
    d, m, T = 8, 16, 120
    A, theta_true, y = _c2_problem(d, m, seed=3)
    rng = np.random.default_rng(6)
    theta0 = theta_true + 0.1 * rng.standard_normal((32, d))

    def run(n, off, th0):
        e = eng_mod.Engine(n, d, seed=7, chain_offset=off)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level(0, A, y, 0, 0.01)
        e.set_proposal(2, 1e-2 * np.eye(d), t0=20, period=20)
        e.init(th0)
        out = e.run_host(T)
        e.close()
        return out

    full = run(32, 0, theta0)
    lo, hi = run(16, 0, theta0[:16]), run(16, 16, theta0[16:])
    for k in range(3):
        assert np.array_equal(full[k][:, :16], lo[k]) and np.array_equal(full[k][:, 16:], hi[k])


def test_full_size_properties(eng_mod):
    """BASELINE config 2 at full size (4096 chains, d=64, m=1024, AM): size-independent invariants."""
    import torch

    d, m, N, T = 64, 1024, 4096, 300
    A, theta_true, y = _c2_problem(d, m)
    e = eng_mod.Engine(N, d, seed=2026)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.init(None)  # theta0 ~ prior from stream 2
    th0, st0 = e.current()
    params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    stats = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    e.run(T, params, stats, acc)
    P, S, Acc = params.cpu().numpy(), stats.cpu().numpy(), acc.cpu().numpy()
    assert np.isfinite(P).all() and np.isfinite(S).all()
    assert np.array_equal(S[:, :, 2], S[:, :, 0] + S[:, :, 1])
    prev = np.concatenate([th0[None], P[:-1]], axis=0)
    rej = Acc == 0
    assert np.array_equal(P[rej], prev[rej]), "a rejected step changed the state"
    assert (np.abs(P[~rej] - prev[~rej]).max(axis=-1) > 0).all(), "an accepted step left the state unchanged"
    # recorded log-densities are those of the recorded states (checked with the oracle on a sample)
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    for t in (0, 150, T - 1):
        idx = np.random.default_rng(t).choice(N, 256, replace=False)
        lp, ll, _ = lvl.evaluate(P[t, idx])
        np.testing.assert_allclose(S[t, idx, 2], lp + ll, rtol=RTOL)
    # running moments == moments of the recorded history (sd, eps scaling of utils.py:117-122)
    ps = e.proposal_state(want_am=True)
    hist = np.concatenate([th0[None], P], axis=0)[:, :64]  # first 64 chains
    sd = min(1.0, 2.4 ** 2 / d)
    np.testing.assert_allclose(ps["am_mu"][:64], hist.mean(axis=0), rtol=1e-9, atol=1e-12)
    for c in range(4):
        cov = sd * (np.cov(hist[:, c].T) + 1e-6 * np.eye(d))
        np.testing.assert_allclose(ps["am_sigma"][c], cov, rtol=1e-6, atol=1e-9)
    assert not e.flags().any()
    assert ps["t"] == T
    rate = Acc.mean()
    assert 0.05 < rate < 0.9, rate
    e.close()


def test_sample_api_on_device(eng_mod):
    """tinyda_amd.sample end to end on the GPU: MH / DA / MLDA result dicts with the reference's keys and lengths."""
    import scipy.stats as st

    import tinyda_amd as tda

    rng = np.random.default_rng(3)
    d = 5
    truth = rng.standard_normal(d)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    posts = []
    for m in (8, 16, 30):
        A = rng.standard_normal((m, d)) / np.sqrt(d)
        posts.append(tda.Posterior(prior, tda.GaussianLogLike(A @ truth + 0.1 * rng.standard_normal(m), 0.01 * np.eye(m)),
                                   tda.LinearModel(A)))
    A2 = posts[2].model.A
    cov2 = np.linalg.inv(A2.T @ A2 / 0.01 + np.eye(d))
    mean2 = cov2 @ (A2.T @ posts[2].likelihood.data / 0.01)
    start = [mean2 + 0.01 * rng.standard_normal(d) for _ in range(24)]
    res = tda.sample(posts[2], tda.AdaptiveMetropolis(1e-3 * np.eye(d), t0=50, period=50), 400, n_chains=24, seed=5,
                     initial_parameters=start)
    assert res["sampler"] == "MH" and res["backend"] == "hip" and res["iterations"] == 401 and len(res["chain_23"]) == 401
    link = res["chain_3"][400]
    assert isinstance(link, tda.Link) and link.model_output.shape == (30,)
    ref = posts[2].create_link(link.parameters)
    np.testing.assert_allclose([link.prior, link.likelihood], [ref.prior, ref.likelihood], rtol=1e-10)
    s = tda.get_samples(res, burnin=200)
    assert s["chain_0"].shape == (201, d)
    pooled = np.concatenate([s["chain_%d" % i] for i in range(24)])
    # loose sanity check of the sampled posterior against the conjugate answer (parity is tested elsewhere)
    assert np.all(np.abs(pooled.mean(0) - mean2) < 1.0 * np.sqrt(np.diag(cov2)))
    assert np.all(np.abs(pooled.std(0) / np.sqrt(np.diag(cov2)) - 1) < 0.5)

    da = tda.sample(posts[1:], tda.CrankNicolson(scaling=0.2), 50, n_chains=8, subsampling_rate=4, seed=6)
    assert da["sampler"] == "DA" and da["iterations"] == 51 and da["subchain_length"] == 4
    assert len(da["chain_fine_0"]) == 51 and len(da["chain_coarse_7"]) == 200
    assert tda.get_samples(da, level="coarse")["chain_0"].shape == (200, d)
    assert tda.sample(posts[1:], tda.CrankNicolson(), 5, n_chains=2, subchain_length=2, store_coarse_chain=False, seed=1)["chain_coarse_1"] is None

    ml = tda.sample(posts, tda.AdaptiveMetropolis(1e-2 * np.eye(d), t0=20, period=20), 30, n_chains=8, subchain_length=[3, 2], seed=7)
    assert ml["sampler"] == "MLDA" and ml["levels"] == 3 and ml["subchain_lengths"] == [3, 2]
    assert len(ml["chain_l2_0"]) == 31 and len(ml["chain_l1_0"]) == 60 and len(ml["chain_l0_0"]) == 180
    assert tda.get_samples(ml, level=1)["iterations"] == 60


def test_pooled_adaptive_metropolis_extension(eng_mod):
    """Extension: one AM covariance pooled over all chains (and, under a process group, all GPUs): the device moment
    reduction equals NumPy's, and the sampler recovers the conjugate posterior covariance."""
    import torch

    from tinyda_amd.distributed import PooledAdaptiveMetropolis

    d, m, N, T = 16, 64, 512, 600
    rng = np.random.default_rng(8)
    A = rng.standard_normal((m, d)) / 4
    truth = rng.standard_normal(d)
    y = A @ truth + 0.2 * rng.standard_normal(m)
    cov_post = np.linalg.inv(A.T @ A / 0.04 + np.eye(d))
    mean_post = cov_post @ (A.T @ y / 0.04)
    e = eng_mod.Engine(N, d, seed=21)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.04)
    e.set_proposal(0, 1e-3 * np.eye(d))
    e.init(mean_post + 0.05 * rng.standard_normal((N, d)))
    params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    pam = PooledAdaptiveMetropolis(e, 1e-3 * np.eye(d), t0=100, period=100)
    pam.run(T, params, None, acc)
    P = params.cpu().numpy()
    s = pam.sums.cpu().numpy()
    flat = P.reshape(-1, d)
    np.testing.assert_allclose(s[0], flat.shape[0])
    np.testing.assert_allclose(s[1:1 + d], flat.sum(0), rtol=1e-11)
    np.testing.assert_allclose(s[1 + d:].reshape(d, d), flat.T @ flat, rtol=1e-11)
    tail = P[300:].reshape(-1, d)
    np.testing.assert_allclose(tail.mean(0), mean_post, atol=4 * np.sqrt(np.diag(cov_post)).max() / np.sqrt(200))
    ratio = np.diag(np.cov(tail.T)) / np.diag(cov_post)
    assert np.all((ratio > 0.7) & (ratio < 1.4)), ratio
    np.testing.assert_allclose(e.proposal_state()["C"][0], pam.C, rtol=1e-10)
    assert 0.1 < acc[300:].float().mean().item() < 0.6
    e.close()
