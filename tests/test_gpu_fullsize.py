"""BASELINE configurations 3-5 at full chain counts: size-independent invariants of the device records, plus
spot re-evaluation of recorded states with the oracle."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()
    return engine


def _levels(ms, d=64, seed=2, sigma=0.1):
    rng = np.random.default_rng(seed)
    truth = rng.standard_normal(d)
    return [(A, A @ truth + sigma * rng.standard_normal(m)) for m in ms for A in [rng.standard_normal((m, d)) / 8]]


def _check_level_records(P, S, Acc, th_init, level, n_check, seed):
    assert np.isfinite(P).all() and np.isfinite(S).all()
    assert np.array_equal(S[:, :, 2], S[:, :, 0] + S[:, :, 1])
    rng = np.random.default_rng(seed)
    for t in rng.choice(P.shape[0], 3, replace=False):
        idx = rng.choice(P.shape[1], n_check, replace=False)
        lp, ll, _ = level.evaluate(P[t, idx])
        np.testing.assert_allclose(S[t, idx, 2], lp + ll, rtol=1e-10)


def test_c3_delayed_acceptance_full_size(eng_mod):
    """config 3: 2-level DA, coarse/fine 256/2048 obs, pCN, subsampling 10, 4096 chains."""
    d, N, L, n_fine = 64, 4096, 10, 20
    lv = _levels((256, 2048))
    e = eng_mod.Engine(N, d, seed=3, n_levels=2)
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(1, None, scaling=0.02)
    e.set_subchains([L])
    e.init(None)
    th0, _ = e.level_state(1)
    outs = e.run_levels_host(n_fine)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(A, y, "iso", 0.01, prior) for A, y in lv]
    (Pc, Sc, Ac), (Pf, Sf, Af) = outs
    assert Pc.shape == (n_fine * L, N, d) and Pf.shape == (n_fine, N, d)
    _check_level_records(Pc, Sc, Ac, th0, levels[0], 128, 1)
    _check_level_records(Pf, Sf, Af, th0, levels[1], 128, 2)
    prev_f = np.concatenate([th0[None], Pf[:-1]])
    assert np.array_equal(Pf[Af == 0], prev_f[Af == 0]), "a rejected fine step changed the fine state"
    # skip-eval rule: a fine step can only accept if its coarse subchain accepted something (chain.py:357-364)
    any_coarse = Ac.reshape(n_fine, L, N).any(axis=1)
    assert not (Af.astype(bool) & ~any_coarse).any()
    # an accepted fine state is the last coarse state of its subchain
    last_coarse = Pc.reshape(n_fine, L, N, d)[:, -1]
    acc = Af.astype(bool)
    assert np.array_equal(Pf[acc], last_coarse[acc])
    # after a fine rejection the coarse chain restarts from the fine state (chain.py:394-396)
    first_coarse_next = Pc.reshape(n_fine, L, N, d)[1:, 0]
    first_acc_next = Ac.reshape(n_fine, L, N)[1:, 0].astype(bool)
    same = ~first_acc_next  # coarse step rejected -> state = restart point = current fine state
    assert np.array_equal(first_coarse_next[same], Pf[:-1][same])
    assert 0.05 < Ac.mean() < 0.95
    e.close()


def test_c5_mlda_full_size(eng_mod):
    """config 5 (literal variant): 3-level MLDA 128/512/2048 obs, AdaptiveMetropolis, subchains [5, 3], 4096 chains."""
    d, N, sl, n_fine = 64, 4096, [5, 3], 8
    lv = _levels((128, 512, 2048), seed=5)
    e = eng_mod.Engine(N, d, seed=4, n_levels=3)
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=30, period=30)
    e.set_subchains(sl)
    e.init(None)
    th0, _ = e.level_state(2)
    outs = e.run_levels_host(n_fine)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(A, y, "iso", 0.01, prior) for A, y in lv]
    rows = e.rows_per_level(n_fine)
    assert rows == [120, 24, 8]
    for k in range(3):
        assert outs[k][0].shape[0] == rows[k]
        _check_level_records(outs[k][0], outs[k][1], outs[k][2], th0, levels[k], 96, 10 + k)
    A1 = outs[1][2].reshape(n_fine, 3, N).astype(bool)
    A0 = outs[0][2].reshape(n_fine, 3, 5, N).astype(bool)
    assert not (A1 & ~A0.any(axis=2)).any(), "level 1 accepted although its level-0 subchain accepted nothing"
    assert not (outs[2][2].astype(bool) & ~A1.any(axis=1)).any()
    acc2 = outs[2][2].astype(bool)
    last1 = outs[1][0].reshape(n_fine, 3, N, d)[:, -1]
    assert np.array_equal(outs[2][0][acc2], last1[acc2])
    assert not e.flags().any()
    e.close()


def test_c4_dream_full_size(eng_mod):
    """config 4 (one GPU's share): DREAM with a shared archive on the 32-dim Rosenbrock chain, 8192 chains."""
    d, N, T, M0, K = 32, 8192, 64, 320, 16
    e = eng_mod.Engine(N, d, seed=8)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
    e.set_proposal_dreamz(M0, delta=1, nCR=3, adaptive=True, period=32, shared=True, sync_every=K, capacity=M0 + T * N)
    e.set_archive(None)
    e.init(None)
    th0, _ = e.current()
    P, S, Acc = e.run_host(T)
    assert np.isfinite(P).all() and np.isfinite(S).all()
    level = orc.RosenbrockLevel(orc.MVNPrior(np.zeros(d), np.eye(d)))
    for t in (0, 31, T - 1):
        idx = np.random.default_rng(t).choice(N, 256, replace=False)
        lp, ll, _ = level.evaluate(P[t, idx])
        np.testing.assert_allclose(S[t, idx, 2], lp + ll, rtol=1e-10)
    prev = np.concatenate([th0[None], P[:-1]])
    assert np.array_equal(P[Acc == 0], prev[Acc == 0])
    # DREAM jumps only move the crossover subspace: accepted moves differ from the previous state in >= 1 coordinate
    moved = (P != prev).any(axis=-1)
    assert np.array_equal(moved, Acc.astype(bool))
    st = e.dreamz_state()
    assert st["archive_rows"] == M0 + T * N
    np.testing.assert_allclose(st["pCR"].sum(axis=1), 1.0, rtol=1e-12)
    e.close()


@pytest.mark.parametrize("N,m", [(4096, 128), (1024, 256)], ids=["4096x128", "1024x256"])
def test_c5_error_model_full_size(eng_mod, N, m):
    """SURVEY's C5 with the state-independent error model: 4096 chains, d = 64, three linear levels with a common output
    dimension of 128, AdaptiveMetropolis, subchains [5, 3] -- and the same at 256 outputs (round 5: k_aem_refresh_big; 1024 chains,
    the host forms every chain's 256 x 256 inverse for the check).  Invariants: the finest records carry the finest posterior
    (re-evaluated by the oracle), every chain's (Sigma_e + Sigma_bias)^-1 is finite, symmetric and positive on the
    diagonal, biases are finite, the levels move."""
    d, n_fine = 64, 4
    rng = np.random.default_rng(6)
    truth = rng.standard_normal(d)
    Af = rng.standard_normal((m, d)) / 8
    y = Af @ truth + 0.1 * rng.standard_normal(m)
    As = [Af + 0.02 * (2 - k) * rng.standard_normal((m, d)) / 8 for k in range(3)]
    e = eng_mod.Engine(N, d, seed=10, n_levels=3)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(3):
        e.set_level(k, As[k], y, 3 if k < 2 else 0, 0.01 * np.eye(m) if k < 2 else 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.set_subchains([5, 3])
    e.set_error_model("state-independent")
    e.init(truth + 0.02 * rng.standard_normal((N, d)))
    outs = e.run_levels_host(n_fine)
    bias, P = e.error_model_state(0, m)
    e.close()
    Pf, Sf, Af_ = outs[2]
    assert np.isfinite(Pf).all() and np.isfinite(Sf).all() and np.isfinite(bias).all() and np.isfinite(P).all()
    lvl = orc.LinearGaussianLevel(As[2], y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    idx = rng.choice(N, 64, replace=False)
    lp, ll, _ = lvl.evaluate(Pf[-1, idx])
    np.testing.assert_allclose(Sf[-1, idx, 2], lp + ll, rtol=1e-10)
    assert np.allclose(P, np.swapaxes(P, 1, 2), rtol=1e-9, atol=1e-9) and (np.einsum("cii->ci", P) > 0).all()
    assert 0.02 < outs[0][2].mean() < 0.98 and 0.02 < Af_.mean() <= 1.0


def test_c5_diagonal_error_model_full_size(eng_mod):
    """C5 with the diagonal error model at an output dimension the dense model cannot hold: 4096 chains, d = 64, three linear
    levels of 1024 outputs, AdaptiveMetropolis, subchains [5, 3].  Invariants: finest records carry the finest posterior
    (oracle re-evaluation), biases finite and of the size of the model discrepancy, every level moves, the skip rule holds."""
    N, d, m, n_fine = 4096, 64, 1024, 3
    rng = np.random.default_rng(16)
    truth = rng.standard_normal(d)
    Af = rng.standard_normal((m, d)) / 8
    y = Af @ truth + 0.1 * rng.standard_normal(m)
    As = [Af + 0.02 * (2 - k) * rng.standard_normal((m, d)) / 8 for k in range(3)]
    e = eng_mod.Engine(N, d, seed=12, n_levels=3)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(3):
        e.set_level(k, As[k], y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.set_subchains([5, 3])
    e.set_error_model("state-independent-diagonal")
    th0 = truth + 0.02 * rng.standard_normal((N, d))
    e.init(th0)
    outs = e.run_levels_host(n_fine)
    bias0, _ = e.error_model_state(0, m, covariance=False)
    bias1, _ = e.error_model_state(1, m, covariance=False)
    assert not e.flags().any()
    e.close()
    Pf, Sf, Af_ = outs[2]
    assert all(np.isfinite(o[0]).all() and np.isfinite(o[1]).all() for o in outs)
    lvl = orc.LinearGaussianLevel(As[2], y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    idx = rng.choice(N, 64, replace=False)
    lp, ll, _ = lvl.evaluate(Pf[-1, idx])
    np.testing.assert_allclose(Sf[-1, idx, 2], lp + ll, rtol=1e-10)
    # the bias of level k estimates F_{k+1}(theta) - F_k(theta) at the states the chains visit (chain.py:1016-1040)
    for k, bias in ((0, bias0), (1, bias1)):
        assert np.isfinite(bias).all()
        gap = (As[k + 1] - As[k]) @ truth
        assert np.abs(bias.mean(axis=0) - gap).max() < 0.2 * np.abs(gap).max() + 0.05
    A1 = outs[1][2].reshape(n_fine, 3, N).astype(bool)
    A0 = outs[0][2].reshape(n_fine, 3, 5, N).astype(bool)
    assert not (A1 & ~A0.any(axis=2)).any()
    assert not (Af_.astype(bool) & ~A1.any(axis=1)).any()
    assert 0.02 < outs[0][2].mean() < 0.98 and 0.02 < Af_.mean() <= 1.0


def test_mlda_dreamz_full_size(eng_mod):
    """DREAM(Z) below a three-level hierarchy (the reference's MLDA notebook, cells 20-23) at 4096 chains: per-chain archives
    grow by one row per base step, records of every level re-evaluate with the oracle, the skip rule holds."""
    d, N, sl, n_fine, M0 = 16, 4096, [4, 2], 6, 48
    rng = np.random.default_rng(26)
    truth = rng.standard_normal(d) * 0.5
    lv = []
    for m in (32, 64, 128):
        A = rng.standard_normal((m, d)) / 4
        lv.append((A, A @ truth + 0.1 * rng.standard_normal(m)))
    base_steps = n_fine * sl[0] * sl[1]
    e = eng_mod.Engine(N, d, seed=14, n_levels=3)
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, 0.01)
    e.set_proposal_dreamz(M0, delta=1, nCR=3, adaptive=True, period=8, capacity=M0 + base_steps)
    e.set_subchains(sl)
    e.set_archive(rng.standard_normal((N, M0, d)))
    e.init(np.tile(truth, (N, 1)) + 0.05 * rng.standard_normal((N, d)))
    th0, _ = e.level_state(2)
    outs = e.run_levels_host(n_fine)
    st = e.dreamz_state()
    e.close()
    assert st["archive_rows"] == M0 + base_steps
    np.testing.assert_allclose(st["pCR"].sum(axis=1), 1.0, rtol=1e-12)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(A, y, "iso", 0.01, prior) for A, y in lv]
    for k in range(3):
        _check_level_records(outs[k][0], outs[k][1], outs[k][2], th0, levels[k], 96, 30 + k)
    A1 = outs[1][2].reshape(n_fine, sl[1], N).astype(bool)
    A0 = outs[0][2].reshape(n_fine, sl[1], sl[0], N).astype(bool)
    assert not (A1 & ~A0.any(axis=2)).any()
    assert not (outs[2][2].astype(bool) & ~A1.any(axis=1)).any()
    assert 0.02 < outs[0][2].mean() < 0.98


def test_128_parameters_full_size(eng_mod):
    """round 5: 4096 chains at 128 parameters, 1024 observations, AdaptiveMetropolis across two covariance swaps, and Delayed
    Acceptance 256 / 2048 under pCN -- the same invariants as at 64 parameters: finite records, posterior = prior + likelihood,
    recorded states re-evaluated by the oracle to 1e-10, rejected steps leave the state alone, every chain's swapped-in covariance
    symmetric and positive on the diagonal"""
    d, N, T = 128, 4096, 220
    (A, y), = _levels((1024,), d=d)
    e = eng_mod.Engine(N, d, seed=5)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.init(None)
    th0, _ = e.current()
    P, S, Acc = e.run_host(T)
    st = e.proposal_state(want_am=True)
    e.close()
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.01, orc.MVNPrior(np.zeros(d), np.eye(d)))
    _check_level_records(P, S, Acc, th0, lvl, 96, 3)
    prev = np.concatenate([th0[None], P[:-1]])
    assert np.array_equal(P[Acc == 0], prev[Acc == 0]), "a rejected step changed the state"
    assert 0.02 < Acc.mean() < 0.98
    C = np.asarray(st["C"])
    assert C.shape == (N, d, d) and np.isfinite(C).all() and np.allclose(C, np.swapaxes(C, 1, 2), rtol=1e-9, atol=1e-14) and (np.einsum("cii->ci", C) > 0).all()
    lv = _levels((256, 2048), d=d)
    L, n_fine = 10, 12
    e = eng_mod.Engine(N, d, seed=6, n_levels=2)
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (Ak, yk) in enumerate(lv):
        e.set_level(k, Ak, yk, 0, 0.01)
    e.set_proposal(1, None, scaling=0.02)
    e.set_subchains([L])
    e.init(None)
    th0, _ = e.level_state(1)
    (Pc, Sc, Ac), (Pf, Sf, Af) = e.run_levels_host(n_fine)
    e.close()
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(Ak, yk, "iso", 0.01, prior) for Ak, yk in lv]
    _check_level_records(Pc, Sc, Ac, th0, levels[0], 96, 4)
    _check_level_records(Pf, Sf, Af, th0, levels[1], 96, 5)
    any_coarse = Ac.reshape(n_fine, L, N).any(axis=1)
    assert not (Af.astype(bool) & ~any_coarse).any()
    assert 0.0 < Af.mean() <= 1.0
