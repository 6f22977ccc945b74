"""OperatorWeightedCrankNicolson (tinyDA/proposal.py:515-605): host class and device path against traces produced by the
reference (tests/golden/g13_owcn*.npz; the oracle is pinned on them in tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import scipy.stats as st

from oracle import tinyda_oracle as orc


def _level(g):
    return orc.LinearGaussianLevel(g["A"], g["data"], "iso", float(g["noise_var"]), orc.MVNPrior(g["prior_mean"], g["prior_cov"]))


def _posterior(g):
    import tinyda_amd as tda

    return tda.Posterior(st.multivariate_normal(g["prior_mean"], g["prior_cov"]),
                         tda.GaussianLogLike(g["data"], float(g["noise_var"]) * np.eye(len(g["data"]))), tda.LinearModel(g["A"]))


@pytest.mark.parametrize("name", ["g13_owcn", "g13_owcn_adaptive"])
def test_host_class_replays_reference_chain(golden, name, monkeypatch):
    """The host protocol (setup_proposal / make_proposal / get_acceptance / adapt) on the reference's variates."""
    import tinyda_amd as tda

    g = golden(name)
    post = _posterior(g)
    for c in range(2):
        prop = tda.OperatorWeightedCrankNicolson(g["B"], scaling=float(g["scaling0"]), adaptive=bool(g["adaptive"]),
                                                 gamma=float(g["gamma"]), period=int(g["period"]))
        prop.setup_proposal(parameters=g["theta0"][c], posterior=post)
        zs = iter(g["z"][c])
        monkeypatch.setattr(np.random, "standard_normal", lambda n: next(zs))
        link = post.create_link(g["theta0"][c])
        accepted = []
        for s in range(g["z"].shape[1]):
            cand = post.create_link(prop.make_proposal(link))
            acc = g["u"][c, s] < prop.get_acceptance(cand, link)
            if acc:
                link = cand
            accepted.append(acc)
            prop.adapt(parameters=link.parameters, accepted=accepted)
            assert acc == bool(g["accepted"][c, s + 1]), (c, s)
            np.testing.assert_allclose(link.posterior, g["logpost"][c, s + 1], rtol=1e-10)
        np.testing.assert_allclose(prop.scaling, g["scaling_hist"][c, -1], rtol=1e-12)


def test_lowering_rules(golden):
    import tinyda_amd as tda
    from tinyda_amd import api

    g = golden("g13_owcn")
    post = _posterior(g)
    fixed = tda.OperatorWeightedCrankNicolson(g["B"], scaling=0.4)
    plan = api._device_plan([post], fixed)
    assert plan is not None and plan[1]["kind"] == 5
    S, Nop = orc.owcn_operators(g["B"], 0.4)
    np.testing.assert_allclose(plan[1]["state_operator"], S, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(plan[1]["noise_operator"], Nop, rtol=1e-13, atol=1e-15)
    # adaptive: every chain has its own operators; for a symmetric B the engine works them out from B's spectrum
    ad = api._device_plan([post], tda.OperatorWeightedCrankNicolson(g["B"], scaling=0.4, adaptive=True, period=30))
    assert ad is not None and ad[1]["adaptive"] and ad[1]["scaling"] == 0.4
    V, lam = ad[1]["spectrum"]
    np.testing.assert_allclose(V @ np.diag(np.sqrt(1.0 - 0.4 * lam)) @ V.T, S, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(V @ np.diag(np.sqrt(0.4 * lam)) @ V.T, Nop, rtol=1e-12, atol=1e-14)
    skew = g["B"] + 1e-3 * np.triu(np.ones_like(g["B"]), 1)
    assert api._device_plan([post], tda.OperatorWeightedCrankNicolson(skew, scaling=0.4, adaptive=True)) is None  # host protocol
    assert api._device_plan([post, post], fixed) is None


@pytest.mark.gpu
def test_device_replays_reference_chain(golden):
    from tinyda_amd.engine import Engine

    g = golden("g13_owcn")
    N, T1, d = g["theta"].shape
    S, Nop = orc.owcn_operators(g["B"], float(g["scaling0"]))
    e = Engine(N, d, seed=1)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
    e.set_proposal(5, None, state_operator=S, noise_operator=Nop)
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    params, stats, acc = e.run_host(T1 - 1)
    e.close()
    assert np.array_equal(acc, np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(params, np.swapaxes(g["theta"][:, 1:], 0, 1), rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_device_replays_the_adaptive_reference_chain(golden):
    """OperatorWeightedCrankNicolson(adaptive=True) (proposal.py:582-590): every chain recomputes its operators from its own
    scaling each period; the engine gets B's spectrum and does that per chain (k_mh_steps PX = 3).  Against tinyDA's own trace:
    accept flags, log-posteriors, states and the adapted scalings."""
    from tinyda_amd.engine import Engine

    g = golden("g13_owcn_adaptive")
    N, T1, d = g["theta"].shape
    lam, V = np.linalg.eigh(g["B"])
    for block in (0, 17):
        e = Engine(N, d, seed=1, block_steps=block)
        e.set_prior(g["prior_mean"], g["prior_cov"])
        e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
        e.set_proposal(5, None, scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]), period=int(g["period"]), spectrum=(V, lam))
        e.init(g["theta0"])
        e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
        params, stats, acc = e.run_host(T1 - 1)
        sc = e.proposal_state_scaling()
        e.close()
        assert np.array_equal(acc, np.swapaxes(g["accepted"][:, 1:], 0, 1))
        np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
        np.testing.assert_allclose(params, np.swapaxes(g["theta"][:, 1:], 0, 1), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(sc, g["scaling_hist"][:, -1], rtol=1e-12)
    with pytest.raises(Exception):  # adaptive without the spectrum: refused at init, never run with one operator for all chains
        e = Engine(N, d, seed=1)
        try:
            e.set_prior(g["prior_mean"], g["prior_cov"])
            e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
            S, Nop = orc.owcn_operators(g["B"], float(g["scaling0"]))
            e.set_proposal(5, None, adaptive=True, state_operator=S, noise_operator=Nop)
            e.init(g["theta0"])
        finally:
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d,m,N,noise,sc", [(64, 256, 40, "iso", 0.004), (7, 20, 21, "diag", 0.05), (24, 48, 16, "iso", 0.02)])
def test_device_adaptive_forward_mode_vs_oracle(d, m, N, noise, sc):
    """Philox mode with adaptation over several periods incl. a split run and a checkpoint, every padded dimension class"""
    from tinyda_amd.engine import Engine

    rng = np.random.default_rng(100 + d)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = rng.standard_normal(d)
    y = A @ truth + 0.2 * rng.standard_normal(m)
    R = rng.standard_normal((d, d)) / np.sqrt(d)
    pc = R @ R.T + 0.5 * np.eye(d)
    Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    B = Q @ np.diag(np.linspace(0.02, 0.5, d)) @ Q.T
    B = 0.5 * (B + B.T)
    lam, V = np.linalg.eigh(B)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    nz = 0.04 if noise == "iso" else 0.02 + 0.04 * rng.random(m)
    T, period = 170, 25
    e = Engine(N, d, seed=78, chain_offset=5)
    e.set_prior(np.zeros(d), pc)
    e.set_level(0, A, y, 0 if noise == "iso" else 1, nz)
    e.set_proposal(5, None, scaling=sc, adaptive=True, gamma=1.05, period=period, spectrum=(V, lam))
    e.init(theta0)
    z, u = e.set_export(T)
    p1, s1, a1 = e.run_host(60)
    blob = e.get_state()
    p2, s2, a2 = e.run_host(T - 60)
    sc_end = e.proposal_state_scaling()
    e.set_state(blob)
    e.close()
    stats, acc = np.concatenate([s1, s2]), np.concatenate([a1, a2])
    lvl = orc.LinearGaussianLevel(A, y, noise, nz, orc.MVNPrior(np.zeros(d), pc))
    ref = orc.run_mh(lvl, dict(kind="owcn", B=B, scaling=sc, adaptive=True, gamma=1.05, period=period), theta0, np.swapaxes(z, 0, 1),
                     np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(sc_end, ref["scaling"], rtol=1e-12)
    assert 0.02 < acc.mean() < 0.98 and np.ptp(sc_end) > 0  # the chains did adapt, each for itself


@pytest.mark.gpu
@pytest.mark.parametrize("d,m,N,block,noise,sc", [(64, 256, 40, 0, "iso", 0.002), (7, 20, 21, 33, "diag", 0.05), (24, 48, 16, 0, "dense", 0.01)])
def test_device_forward_mode_vs_oracle(d, m, N, block, noise, sc):
    """Philox mode incl. split runs and a checkpoint, every padded dimension class and noise kind."""
    from tinyda_amd.engine import Engine

    rng = np.random.default_rng(d)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    truth = rng.standard_normal(d)
    y = A @ truth + 0.2 * rng.standard_normal(m)
    R = rng.standard_normal((d, d)) / np.sqrt(d)
    pc = R @ R.T + 0.5 * np.eye(d)
    Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    B = Q @ np.diag(np.linspace(0.02, 0.5, d)) @ Q.T
    S, Nop = orc.owcn_operators(B, sc)
    theta0 = truth + 0.05 * rng.standard_normal((N, d))
    if noise == "iso":
        kind, nz, onz = 0, 0.04, 0.04
    elif noise == "diag":
        nz = 0.02 + 0.04 * rng.random(m)
        kind, onz = 1, nz
    else:
        Ln = 0.2 * np.eye(m) + 0.02 * np.tril(rng.standard_normal((m, m)))
        nz = Ln @ Ln.T
        kind, onz = 2, nz
    T = 150
    e = Engine(N, d, seed=77, chain_offset=3, block_steps=block)
    e.set_prior(np.zeros(d), pc)
    e.set_level(0, A, y, kind, nz)
    e.set_proposal(5, None, state_operator=S, noise_operator=Nop)
    e.init(theta0)
    z, u = e.set_export(T)
    p1, s1, a1 = e.run_host(60)
    blob = e.get_state()
    p2, s2, a2 = e.run_host(T - 60)
    e.set_state(blob)
    e.close()
    stats, acc = np.concatenate([s1, s2]), np.concatenate([a1, a2])
    lvl = orc.LinearGaussianLevel(A, y, noise, onz, orc.MVNPrior(np.zeros(d), pc))
    ref = orc.run_mh(lvl, dict(kind="owcn", B=B, scaling=sc), theta0, np.swapaxes(z, 0, 1), np.swapaxes(u, 0, 1))
    assert np.array_equal(acc, np.swapaxes(ref["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(ref["logpost"][:, 1:], 0, 1), rtol=1e-10)
    assert 0.02 < acc.mean() < 0.98


@pytest.mark.gpu
def test_sample_api_owcn(golden):
    import tinyda_amd as tda

    g = golden("g13_owcn")
    post = _posterior(g)
    res = tda.sample(post, tda.OperatorWeightedCrankNicolson(g["B"], scaling=0.4), 200, n_chains=8, seed=4)
    assert res["sampler"] == "MH" and res.get("backend", "hip") != "host"
    link = res["chain_5"][-1]
    assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)
    ad = tda.sample(post, tda.OperatorWeightedCrankNicolson(g["B"], scaling=0.4, adaptive=True, period=40), 200, n_chains=8, seed=4)
    assert ad["backend"] == "hip" and np.ptp(ad["proposal_state"]["scaling"]) > 0 and ad["proposal_state"]["k"] == 5
    link = ad["chain_2"][-1]
    assert np.isclose(link.posterior, post.create_link(link.parameters).posterior, rtol=1e-10)
