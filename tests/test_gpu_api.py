"""tda.sample() on the device path: records stay in HBM (lazy DeviceChain views, bulk get_samples), the proposal state is read on
demand, thinning, polled progress, the engine's buffer pool -- and what the call delivers end to end at BASELINE config 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.stats as st

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()  # fail loudly if the HIP library is not built
    return engine


def _problem(d=7, m=33, seed=3):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
    return A, y


def _engine_records(A, y, N, T, seed, prop, theta0=None, thin=1, where="host", splits=None):
    from tinyda_amd.engine import Engine, pinned_empty

    d = A.shape[1]
    e = Engine(N, d, seed=seed)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(**prop)
    e.init(theta0)
    if thin > 1:
        e.set_record_thinning(thin)
    th0, st0 = e.current()
    R = T // thin
    if where == "pinned":
        bufs = pinned_empty((R, N, d)), pinned_empty((R, N, 3)), pinned_empty((R, N), dtype=np.uint8)
    elif where == "device":
        import torch

        bufs = (torch.empty((R, N, d), dtype=torch.float64, device="cuda"), torch.empty((R, N, 3), dtype=torch.float64, device="cuda"),
                torch.empty((R, N), dtype=torch.uint8, device="cuda"))
    else:
        bufs = np.empty((R, N, d)), np.empty((R, N, 3)), np.empty((R, N), dtype=np.uint8)
    t, row = 0, 0
    for n in (splits or [T]):
        k = (t + n) // thin - t // thin
        e.run(n, bufs[0][row:row + k], bufs[1][row:row + k], bufs[2][row:row + k])
        t, row = t + n, row + k
    assert row == R
    state = e.proposal_state(want_am=prop["kind"] == 2)
    e.close()
    if where == "device":
        bufs = tuple(b.cpu().numpy() for b in bufs)
    return th0, st0, bufs, state


def test_sample_keeps_records_on_the_device_and_returns_the_engine_s_records(eng_mod):
    import torch

    import tinyda_amd as tda
    from tinyda_amd.records import DeviceChain

    A, y = _problem()
    d, N, T = A.shape[1], 40, 230
    post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(len(y))), tda.LinearModel(A))
    prop = dict(kind=2, C_=1e-3 * np.eye(d), t0=50, period=50)
    th0, st0, (P, S, Acc), state = _engine_records(A, y, N, T, 11, prop)
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d), t0=50, period=50), T, n_chains=N, seed=11)
    assert res["iterations"] == T + 1 and res["backend"] == "hip"
    ch = res["chain_5"]
    assert isinstance(ch, DeviceChain) and len(ch) == T + 1 and not ch._cache  # nothing fetched yet
    recs = ch._records
    assert recs.on_device and isinstance(recs.parameters, torch.Tensor) and recs.parameters.is_cuda and recs.parameters.shape == (T + 1, N, d)
    # one chain, lazily
    assert np.array_equal(ch.parameters[0], th0[5]) and np.array_equal(ch.parameters[1:], P[:, 5])
    assert np.array_equal(ch.stats[1:], S[:, 5]) and np.array_equal(ch.accepted[1:], Acc[:, 5]) and ch.accepted[0] == 1
    # slices of untouched chains stay lazy and compose
    tail = res["chain_6"][100:][10:]
    assert not res["chain_6"]._cache and len(tail) == T + 1 - 110 and np.array_equal(tail.parameters, P[109:, 6])
    link = res["chain_7"][T]
    ref = post.create_link(link.parameters)
    np.testing.assert_allclose([link.prior, link.likelihood], [ref.prior, ref.likelihood], rtol=1e-10)
    assert np.array_equal(link.parameters, P[-1, 7]) and link.model_output.shape == y.shape
    # all chains in one pass
    for burnin in (0, 31, T + 1, T + 5):
        s = tda.get_samples(res, burnin=burnin)
        assert s["iterations"] == max(T + 1 - burnin, 0) and s["dimension"] == d
        for i in (0, 5, N - 1):
            want = np.concatenate([th0[i][None], P[:, i]])[burnin:]
            assert np.array_equal(s["chain_%d" % i], want)
    res2 = tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d), t0=50, period=50), T, n_chains=N, seed=11)
    big = tda.get_samples(res2, "stats", burnin=17)  # (a second engine out of pooled buffers: the same records again)
    for i in range(N):
        assert np.array_equal(big["chain_%d" % i], np.concatenate([st0[i][None], S[:, i]])[17:])
    assert big["chain_0"].base is not None and big["dimension"] == 3
    # the proposal state is read when asked for, from buffers the engine handed over
    ps = res["proposal_state"]
    assert ps._data is None and set(ps) == {"scaling", "C", "am_mu", "am_sigma", "t", "k"}
    assert ps["t"] == T and np.array_equal(ps["C"], state["C"]) and np.array_equal(ps["am_sigma"], state["am_sigma"])
    assert np.array_equal(ps["am_mu"], state["am_mu"]) and ps._snap is None
    # a random walk has no moments
    g = tda.sample(post, tda.GaussianRandomWalk(1e-3 * np.eye(d), adaptive=True, period=40), 100, n_chains=16, seed=2)
    # the result dict pickles and deep-copies like the reference's plain-Python one (the lazy proposal state wraps a ctypes
    # handle); a copy holds the materialised state and the original keeps working
    import copy
    import pickle

    assert g["proposal_state"]._data is None
    g2 = pickle.loads(pickle.dumps(g))
    g3 = copy.deepcopy(g)
    for other in (g2, g3):
        assert other["proposal_state"]["k"] == 2 and np.array_equal(other["proposal_state"]["scaling"], g["proposal_state"]["scaling"])
        assert np.array_equal(other["chain_3"].parameters, g["chain_3"].parameters)
    assert g["proposal_state"]["am_mu"] is None and g["proposal_state"]["scaling"].shape == (16,) and g["proposal_state"]["k"] == 2
    # to_host(): the records leave HBM, the lazy views keep working
    recs_g = g["chain_0"]._records.to_host()
    assert not recs_g.on_device and not recs_g.parameters.is_cuda and np.array_equal(g["chain_9"].parameters, g2["chain_9"].parameters)


@pytest.mark.parametrize("where", ["host", "pinned", "device"])
def test_record_thinning_keeps_every_kth_iteration(eng_mod, where):
    A, y = _problem(d=9, m=20, seed=5)
    N, T = 21, 257
    rng = np.random.default_rng(1)
    theta0 = rng.standard_normal((N, 9)) * 0.1
    prop = dict(kind=2, C_=1e-3 * np.eye(9), t0=30, period=30, adaptive=True)
    _, _, (P, S, Acc), full_state = _engine_records(A, y, N, T, 4, prop, theta0)
    for thin, splits in ((7, None), (7, [3, 100, 4, 150]), (64, [130, 127]), (300, None)):
        _, _, (p, s_, a), state = _engine_records(A, y, N, T, 4, prop, theta0, thin=thin, where=where, splits=splits)
        keep = np.arange(thin - 1, T, thin)
        assert p.shape[0] == keep.size
        assert np.array_equal(p, P[keep]) and np.array_equal(s_, S[keep]) and np.array_equal(a, Acc[keep])
        assert np.array_equal(state["am_sigma"], full_state["am_sigma"])  # adaptation saw every iteration


def test_sample_thin_and_its_refusals(eng_mod):
    import tinyda_amd as tda

    A, y = _problem()
    d = A.shape[1]
    post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(len(y))), tda.LinearModel(A))
    full = tda.get_samples(tda.sample(post, tda.CrankNicolson(scaling=0.1), 95, n_chains=6, seed=9))
    thinned = tda.sample(post, tda.CrankNicolson(scaling=0.1), 95, n_chains=6, seed=9, thin=10)
    assert thinned["iterations"] == 10 and thinned["thin"] == 10 and len(thinned["chain_2"]) == 10
    got = tda.get_samples(thinned)
    for i in range(6):
        assert np.array_equal(got["chain_%d" % i], full["chain_%d" % i][[0] + list(range(10, 96, 10))])
    with pytest.raises(NotImplementedError):
        tda.sample([post, post], tda.CrankNicolson(), 5, n_chains=2, subchain_length=2, thin=2)
    with pytest.raises(ValueError):
        tda.sample(post, tda.CrankNicolson(), 5, thin=0)


def test_progress_is_polled_without_synchronising(eng_mod, capfd):
    import time

    import torch

    import tinyda_amd as tda
    from tinyda_amd.engine import Engine

    A, y = _problem(d=16, m=64)
    N, T = 512, 1500
    e = Engine(N, 16, seed=1)
    e.set_prior(np.zeros(16), np.eye(16))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-3 * np.eye(16), t0=100, period=100)
    e.init(None)
    with pytest.raises(tda.EngineError):
        e.progress()  # off until asked for
    e.set_progress(True)
    acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    e.run(T, None, None, acc, sync=False)
    seen, t0 = [], time.time()
    while time.time() - t0 < 60:
        done, queued, rate = e.progress()
        seen.append(done)
        if done >= T:
            break
    assert seen[-1] == T and queued == T and sorted(seen) == seen and all(v % 100 == 0 for v in seen)
    e.sync()
    assert abs(rate - acc[-100:].double().mean().item()) < 1e-12  # mean accept flag of the last block
    e.run(250, sync=True)  # no accept buffer: the rate is reported as unknown
    assert e.progress() == (T + 250, T + 250, -1.0)
    e.set_progress(False)
    e.close()
    d = A.shape[1]
    post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(len(y))), tda.LinearModel(A))
    capfd.readouterr()
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d)), 300, n_chains=32, seed=3, force_progress_bar=True)
    err = capfd.readouterr().err
    assert "300/300 iterations" in err and "acceptance" in err
    quiet = tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d)), 300, n_chains=32, seed=3)
    assert np.array_equal(tda.get_samples(res)["chain_31"], tda.get_samples(quiet)["chain_31"])
    da = tda.sample([post, post], tda.CrankNicolson(scaling=0.1), 40, n_chains=16, subchain_length=3, seed=3, force_progress_bar=True)
    assert "40/40 iterations" in capfd.readouterr().err and len(da["chain_fine_0"]) == 41


def test_engine_buffers_are_pooled_and_come_back_zeroed(eng_mod):
    import tinyda_amd as tda
    from tinyda_amd import _lib

    lib = _lib.load()
    assert b"0.5" in lib.tda_version()
    A, y = _problem(d=33, m=70)  # padded to 64 parameters: pad lanes of recycled buffers must read as zero
    d = A.shape[1]
    post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(len(y))), tda.LinearModel(A))
    lib.tda_release_cached_memory()
    runs = [tda.get_samples(tda.sample(post, tda.AdaptiveMetropolis(1e-3 * np.eye(d), t0=20, period=20), 90, n_chains=37, seed=8), "stats")
            for _ in range(3)]
    for r in runs[1:]:
        for i in range(37):
            assert np.array_equal(r["chain_%d" % i], runs[0]["chain_%d" % i])
    assert lib.tda_release_cached_memory() > 0 and lib.tda_release_cached_memory() == 0


def test_sample_end_to_end_rate_at_config2_size():
    """BASELINE configs[1] through tda.sample() in a fresh process that never imports torch itself: first-call overhead under 2 s,
    later calls at >= 1.5e8 proposal evaluations per second end to end (profiles/r03_api.json is this output)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "api_rate.py"), "2000", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(out))
    assert not out["torch_imported_by_script"]
    assert out["first_call_overhead_seconds"] < 2.0, out
    assert out["evals_per_s_end_to_end"] >= 1.5e8, out
    assert out["proposal_state_C_shape"] == [4096, 64, 64]


def test_sample_hierarchy_with_dense_observation_covariance(eng_mod):
    """tda.sample over a Delayed-Acceptance hierarchy whose fine level has a DENSE observation covariance (DefaultGaussianLogLike,
    distributions.py:246-301): lowered to the device since 0.4 (round 3 ran the host protocol, silently)"""
    import warnings

    import tinyda_amd as tda

    rng = np.random.default_rng(5)
    d, m0, m1 = 6, 12, 30
    A0, A1 = rng.standard_normal((m0, d)) / 3, rng.standard_normal((m1, d)) / 3
    truth = rng.standard_normal(d)
    Lc = 0.2 * np.eye(m1) + 0.02 * np.tril(rng.standard_normal((m1, m1)))
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    posts = [tda.Posterior(prior, tda.GaussianLogLike(A0 @ truth, 0.04 * np.eye(m0)), tda.LinearModel(A0)),
             tda.Posterior(prior, tda.GaussianLogLike(A1 @ truth, Lc @ Lc.T), tda.LinearModel(A1))]
    assert type(posts[1].likelihood).__name__ == "DefaultGaussianLogLike"
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        res = tda.sample(posts, tda.CrankNicolson(scaling=0.2), 40, n_chains=24, subchain_length=3, seed=3)
    assert res["sampler"] == "DA" and res["backend"] == "hip"
    for i in (0, 11, 23):
        link = res["chain_fine_%d" % i][-1]
        ref = posts[1].create_link(link.parameters)
        np.testing.assert_allclose([link.prior, link.likelihood], [ref.prior, ref.likelihood], rtol=1e-10)
    acc = np.mean([np.mean(res["chain_fine_%d" % i].accepted[1:]) for i in range(24)])
    assert 0.02 < acc < 0.98


def test_sample_five_level_mlda_runs_on_the_device(eng_mod):
    """tda.sample over FIVE levels (0.5: six at most; 0.4 fell back to the host protocol beyond four): backend hip, the finest links
    carry the finest posterior; with an error model five levels still fall back, announced"""
    import warnings

    import tinyda_amd as tda

    rng = np.random.default_rng(8)
    d, ms = 5, (6, 9, 12, 18, 30)
    base = rng.standard_normal((ms[-1], d)) / 2
    truth = rng.standard_normal(d)
    y = base @ truth + 0.2 * rng.standard_normal(ms[-1])
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    posts = [tda.Posterior(prior, tda.GaussianLogLike(y[:m], 0.04 * np.eye(m)), tda.LinearModel(base[:m])) for m in ms]
    with warnings.catch_warnings():
        warnings.simplefilter("error", tda.HostFallbackWarning)
        res = tda.sample(posts, tda.GaussianRandomWalk(0.01 * np.eye(d)), 12, n_chains=16, subchain_length=[2, 2, 2, 2], seed=5)
    assert res["sampler"] == "MLDA" and res["backend"] == "hip" and res["levels"] == 5
    for i in (0, 7, 15):
        link = res["chain_l4_%d" % i][-1]
        ref = posts[4].create_link(link.parameters)
        np.testing.assert_allclose([link.prior, link.likelihood], [ref.prior, ref.likelihood], rtol=1e-10)
    same = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y[:6], 0.04 * np.eye(6)), tda.LinearModel(base[:6] * (1 + 0.01 * k))) for k in range(4)]
    same.append(tda.Posterior(prior, tda.GaussianLogLike(y[:6], 0.04 * np.eye(6)), tda.LinearModel(base[:6])))
    with pytest.warns(tda.HostFallbackWarning, match="more than 4 levels are lowered without error model"):
        tda.sample(same, tda.GaussianRandomWalk(0.01 * np.eye(d)), 2, n_chains=1, subchain_length=[2, 2, 2, 2], adaptive_error_model="state-independent")
