"""GPU parity of the multi-level engine (Delayed Acceptance, MLDA) against tinyDA's DAChain / MLDAChain golden
traces (replayed variates) and against the oracle on the engine's own Philox stream."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu
RTOL = 1e-10
KIND = {"grw": 0, "pcn": 1, "am": 2}


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()
    return engine


def _prop(g):
    p = {k[5:]: g[k] for k in g.files if k.startswith("prop_")}
    return {k: (str(v) if k == "kind" else (v if v.ndim else v.item())) for k, v in p.items()}


def _engine(eng_mod, g, nl, sl, randomize=False, block=0, seed=11, n_chains=None, theta0=None):
    th0 = g["theta0"] if theta0 is None else theta0
    N, d = th0.shape
    e = eng_mod.Engine(N, d, seed=seed, n_levels=nl, block_steps=block)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    for k in range(nl):
        if "cov0" in g.files:  # per-level covariances: dense (2) / diagonal (1) / isotropic (0), as GaussianLogLike's factory classifies them
            kind, noise = orc.noise_of(g["cov%d" % k])
            e.set_level(k, g["A%d" % k], g["y%d" % k], {"iso": 0, "diag": 1, "dense": 2}[kind], noise)
        else:
            e.set_level(k, g["A%d" % k], g["y%d" % k], 0, float(g["noise_var"]))
    pr = _prop(g)
    kind = KIND[pr["kind"]]
    if kind == 0:
        e.set_proposal(0, pr["C"], scaling=pr["scaling"], adaptive=pr["adaptive"], gamma=pr["gamma"], period=pr["period"])
    elif kind == 1:
        e.set_proposal(1, None, scaling=pr["scaling"], adaptive=pr["adaptive"], gamma=pr["gamma"], period=pr["period"])
    else:
        e.set_proposal(2, pr["C0"], adaptive=pr["adaptive"], gamma=pr["gamma"], period=pr["period"], sd=pr["sd"],
                       epsilon=pr["epsilon"], t0=pr["t0"])
    e.set_subchains(sl, randomize)
    e.init(th0)
    return e


def _check(outs, g, nl, st_init, atol_params=1e-11):
    for k in range(nl):
        params, stats, acc = outs[k]
        ref_acc, ref_lp, ref_ll, ref_th = g["acc%d" % k], g["lp%d" % k], g["ll%d" % k], g["th%d" % k]
        if k == nl - 1:  # the reference's finest chain carries the initial link
            np.testing.assert_allclose(st_init[:, 2], ref_lp[:, 0] + ref_ll[:, 0], rtol=RTOL)
            ref_acc, ref_lp, ref_ll, ref_th = ref_acc[:, 1:], ref_lp[:, 1:], ref_ll[:, 1:], ref_th[:, 1:]
        assert np.array_equal(acc, ref_acc.T), "level %d: %d accept flips" % (k, int((acc != ref_acc.T).sum()))
        np.testing.assert_allclose(stats[:, :, 2], (ref_lp + ref_ll).T, rtol=RTOL)
        np.testing.assert_allclose(stats[:, :, 1], ref_ll.T, rtol=1e-9)
        np.testing.assert_allclose(params, np.swapaxes(ref_th, 0, 1), rtol=1e-9, atol=atol_params)


@pytest.mark.parametrize("name,block", [("g4_da_pcn", 0), ("g4_da_pcn", 7), ("g4_da_grw_adaptive", 0),
                                        ("g4_da_am_random", 0), ("g4_da_pcn_adaptive_c3shape", 0),
                                        # dense observation covariances inside the hierarchy (VERDICT r3 item 7 / missing 3): the fine
                                        # level only, and both levels under a general prior
                                        ("g4_da_pcn_dense_fine", 0), ("g4_da_pcn_dense_fine", 5), ("g4_da_am_dense_both", 0)])
def test_delayed_acceptance_replay(eng_mod, golden, name, block):
    g = golden(name)
    L = int(g["subchain_length"])
    n_fine = g["th1"].shape[1] - 1
    rnd = bool(g["randomize"])
    e = _engine(eng_mod, g, 2, [L], rnd, block)
    e.set_replay(np.swapaxes(g["z"], 0, 1), g["u0"].T)
    e.set_replay_level(1, g["u1"].T)
    if rnd:
        e.set_replay_level(-1, g["ridx"].T)
    _, st1 = e.level_state(1)
    outs = e.run_levels_host(n_fine)
    _check(outs, g, 2, st1)
    np.testing.assert_allclose(e.proposal_state()["scaling"], g["scaling"], rtol=1e-12)
    e.close()


@pytest.mark.parametrize("name,block", [("g5_mlda_am", 0), ("g5_mlda_am", 5), ("g5_mlda_grw_adaptive", 0),
                                        ("g5_mlda_4level", 0), ("g5_mlda_am_dense", 0),
                                        ("g5_mlda_5level", 0), ("g5_mlda_6level", 0)])
def test_mlda_replay(eng_mod, golden, name, block):
    g = golden(name)
    nl = int(g["n_levels"])
    sl = [int(x) for x in g["subchain_lengths"]]
    n_fine = g["th%d" % (nl - 1)].shape[1] - 1
    e = _engine(eng_mod, g, nl, sl, False, block)
    e.set_replay(np.swapaxes(g["z"], 0, 1), g["u0"].T)
    for k in range(1, nl):
        e.set_replay_level(k, g["u%d" % k].T)
    _, stf = e.level_state(nl - 1)
    outs = e.run_levels_host(n_fine)
    # (g5_mlda_am_dense: AdaptiveMetropolis from a ten-sample covariance under dense likelihoods -- 4 of 5760 parameters, all below
    # 0.01 in magnitude, differ by 3e-11: the factor of a near-singular covariance; masks and densities are held to the usual bars)
    _check(outs, g, nl, stf, atol_params=1e-10 if name == "g5_mlda_am_dense" else 1e-11)
    np.testing.assert_allclose(e.proposal_state()["scaling"], g["scaling"], rtol=1e-12)
    e.close()


def _oracle_uniforms(seed, N, rows, sl, randomize_L=None):
    ps = orc.PhiloxStream(seed)
    chains = np.arange(N)
    us = [np.stack([ps.uniform(chains, t, level=k) for t in range(rows[k])], axis=1) for k in range(len(rows))]
    ridx = None
    if randomize_L:
        x0 = np.stack([ps.words(chains.astype(np.uint32), np.uint32(t), np.uint32(3), np.uint32(0))[0] for t in range(rows[1])], axis=1)
        ridx = ((x0.astype(np.uint64) * np.uint64(randomize_L)) >> np.uint64(32)).astype(np.float64) - randomize_L
    return us, ridx


@pytest.mark.parametrize("case", ["da_c3", "da_random", "mlda3", "da_long_data", "mlda5", "mlda6"])
def test_multilevel_philox_forward_vs_oracle(eng_mod, case):
    """Engine on its own Philox stream (normals exported, uniforms / promoted index regenerated bit-exactly by the
    oracle's Philox); BASELINE config-3 / config-5 shapes at reduced chain counts.  da_long_data: a fine level of 9 000
    observations -- its data vector beside the lean kernel's LDS rows exceeds 160 KB, the launcher takes the generic kernel."""
    rng = np.random.default_rng(77)
    if case == "da_long_data":
        d, ms, sl, n_fine, N = 64, (256, 9000), [10], 6, 32
        prop = dict(kind="pcn", scaling=0.02, adaptive=True, gamma=1.01, period=40)
    elif case == "mlda3":
        d, ms, sl, n_fine, N = 64, (128, 512, 2048), [5, 3], 12, 32
        prop = dict(kind="am", C0=1e-4 * np.eye(d), t0=50, period=50)
    elif case == "mlda5":  # five and six levels (round 5): the generic level kernel's k_ml_steps<., 5 | 6> instances
        d, ms, sl, n_fine, N = 64, (64, 96, 160, 320, 640), [3, 2, 2, 2], 6, 32
        prop = dict(kind="pcn", scaling=0.02, adaptive=True, gamma=1.01, period=40)
    elif case == "mlda6":
        d, ms, sl, n_fine, N = 24, (16, 32, 48, 64, 128, 256), [2, 2, 2, 2, 2], 5, 20
        prop = dict(kind="am", C0=1e-3 * np.eye(d), t0=40, period=40)
    else:
        d, ms, sl, n_fine, N = 64, (256, 2048), [10], 20, 48
        prop = dict(kind="pcn", scaling=0.02, adaptive=True, gamma=1.01, period=40) if case == "da_c3" else \
            dict(kind="am", C0=1e-4 * np.eye(d), t0=50, period=50, adaptive=True)
    truth = rng.standard_normal(d)
    As = [rng.standard_normal((m, d)) / 8 for m in ms]
    ys = [A @ truth + 0.1 * rng.standard_normal(len(A)) for A in As]
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    nl = len(ms)
    seed = 31337
    e = eng_mod.Engine(N, d, seed=seed, n_levels=nl)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(nl):
        e.set_level(k, As[k], ys[k], 0, 0.01)
    if prop["kind"] == "pcn":
        e.set_proposal(1, None, scaling=prop["scaling"], adaptive=True, gamma=prop["gamma"], period=prop["period"])
    else:
        e.set_proposal(2, prop["C0"], t0=prop["t0"], period=prop["period"], adaptive=prop.get("adaptive", False))
    e.set_subchains(sl, case == "da_random")
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    us, ridx = _oracle_uniforms(seed, N, rows, sl, sl[0] if case == "da_random" else None)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(As[k], ys[k], "iso", 0.01, prior) for k in range(nl)]
    res, pstate = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, ridx)
    for k in range(nl):
        ref = res[k]
        sk = slice(1, None) if k == nl - 1 else slice(None)
        assert np.array_equal(outs[k][2], ref["accepted"][:, sk].T), "level %d accept masks differ" % k
        np.testing.assert_allclose(outs[k][1][:, :, 2], ref["logpost"][:, sk].T, rtol=RTOL)
    np.testing.assert_allclose(e.proposal_state()["scaling"], pstate.scaling, rtol=1e-12)
    e.close()
