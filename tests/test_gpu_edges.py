"""Edge cases of the device engine: degenerate sizes, ragged tiles, split runs, loud failures."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    from tinyda_amd import _lib, engine

    _lib.load()
    return engine


def _problem(d, m, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    return A, rng.standard_normal(m), rng


@pytest.mark.parametrize("d,m,N", [(1, 1, 1), (1, 5, 3), (2, 1, 17), (5, 17, 33), (64, 16, 15), (63, 31, 16), (33, 2, 31)])
def test_degenerate_and_ragged_shapes(eng_mod, d, m, N):
    """one parameter, one observation, one chain, chain counts that are not a multiple of the 16-chain tile"""
    A, y, rng = _problem(d, m, d * 100 + m)
    T = 40
    theta0 = 0.3 * rng.standard_normal((N, d))
    z = rng.standard_normal((T, N, d))
    u = rng.random((T, N))
    e = eng_mod.Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.5)
    e.set_proposal(2, 0.05 * np.eye(d), t0=10, period=10, adaptive=True)
    e.init(theta0)
    e.set_replay(z, u)
    params, stats, acc = e.run_host(T)
    lvl = orc.LinearGaussianLevel(A, y, "iso", 0.5, orc.MVNPrior(np.zeros(d), np.eye(d)))
    ref = orc.run_mh(lvl, dict(kind="am", C0=0.05 * np.eye(d), t0=10, period=10, adaptive=True), theta0,
                     np.swapaxes(z, 0, 1), u.T)
    assert np.array_equal(acc, ref["accepted"][:, 1:].T)
    np.testing.assert_allclose(stats[:, :, 2], ref["logpost"][:, 1:].T, rtol=1e-10)
    e.close()


def test_zero_and_single_iterations_and_split_runs(eng_mod):
    d, m, N = 7, 11, 20
    A, y, rng = _problem(d, m, 5)

    def make():
        e = eng_mod.Engine(N, d, seed=77, block_steps=13)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level(0, A, y, 1, 0.2 + rng.random(m) * 0 + 0.3)
        e.set_proposal(2, 0.02 * np.eye(d), t0=20, period=20, adaptive=True)
        e.init(np.zeros((N, d)) + 0.1)
        return e

    e = make()
    th_a, st_a = e.current()
    e.run(0)  # no-op
    th_b, st_b = e.current()
    assert np.array_equal(th_a, th_b) and np.array_equal(st_a, st_b)
    whole = e.run_host(200)
    e.close()
    e = make()
    parts = [e.run_host(n) for n in (1, 129, 70)]  # splits that do not line up with blocks or adaptation periods
    for k in range(3):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), whole[k]), "state is not carried across run() calls"
    assert e.proposal_state()["t"] == 200
    e.close()


def test_records_are_optional(eng_mod):
    d, m, N = 4, 6, 16
    A, y, _ = _problem(d, m, 6)
    outs = []
    for want in ((True, True, True), (False, True, False), (False, False, False)):
        e = eng_mod.Engine(N, d, seed=3)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level(0, A, y, 0, 1.0)
        e.set_proposal(2, 0.1 * np.eye(d), t0=5, period=5)  # AM needs the states even when the caller does not
        e.init(np.zeros((N, d)))
        p = np.empty((30, N, d)) if want[0] else None
        s = np.empty((30, N, 3)) if want[1] else None
        a = np.empty((30, N), dtype=np.uint8) if want[2] else None
        e.run(30, p, s, a)
        outs.append((e.current(), s))
        e.close()
    assert np.array_equal(outs[0][0][0], outs[1][0][0]) and np.array_equal(outs[0][0][0], outs[2][0][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_loud_failures(eng_mod):
    from tinyda_amd import EngineError

    with pytest.raises(EngineError, match="dim=129"):  # (0.5: 128 parameters; hierarchies above 64: tests/test_gpu_wide.py)
        eng_mod.Engine(4, 129)
    with pytest.raises(EngineError, match="n_levels"):
        eng_mod.Engine(4, 3, n_levels=7)  # (0.5: six levels)
    e = eng_mod.Engine(4, 3)
    with pytest.raises(EngineError, match="not initialised"):
        e.run(1)
    with pytest.raises(EngineError, match="positive definite"):
        e.set_prior(np.zeros(3), -np.eye(3))
    e.set_prior(np.zeros(3), np.eye(3))
    with pytest.raises(EngineError, match="positive"):
        e.set_level(0, np.eye(3), np.zeros(3), 0, -1.0)
    e.set_level(0, np.eye(3), np.zeros(3), 0, 1.0)
    with pytest.raises(EngineError, match="set_prior and set_proposal"):
        e.init(np.zeros((4, 3)))
    e.set_proposal(0, np.array([[1.0, 2.0, 0], [2.0, 1.0, 0], [0, 0, 1.0]]))  # indefinite proposal covariance
    with pytest.raises(EngineError, match="positive definite"):
        e.init(np.zeros((4, 3)))
    e.set_proposal(0, np.eye(3))
    e.init(np.zeros((4, 3)))
    e.set_replay(np.zeros((2, 4, 3)), np.zeros((2, 4)))
    with pytest.raises(EngineError, match="replay buffer"):
        e.run(3)
    e.close()


def test_nan_posterior_is_rejected_and_overflow_accepts(eng_mod):
    """proposal.py:254-258: NaN proposal posterior -> alpha = 0; exp overflow -> inf -> accept."""
    d, m, N = 2, 2, 16
    e = eng_mod.Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, np.eye(2), np.zeros(2), 0, 1e-300)  # huge likelihood scale: exp overflows towards the mode
    e.set_proposal(0, np.eye(d))
    theta0 = np.full((N, d), 3.0)
    e.init(theta0)
    z = np.zeros((2, N, d))
    z[0] = -1.0  # towards the mode: log-ratio ~ +1e300 -> exp = inf -> accept whatever u is
    z[1, :, 0] = np.nan  # NaN proposal -> rejected
    u = np.full((2, N), 0.999999)
    e.set_replay(z, u)
    params, stats, acc = e.run_host(2)
    assert acc[0].all() and not acc[1].any()
    assert np.array_equal(params[1], params[0])
    e.close()


@pytest.mark.parametrize("kind", ["am", "mlda", "dreamz", "mlda_aem", "mlda_aemd", "mlda_aemd_wide", "da_aem_sd", "da_random",
                                  "dream_shared", "mlda_dreamz"])
def test_checkpoint_resume_is_bitwise(eng_mod, kind):
    """get_state / set_state: a run interrupted at an awkward point (mid period, mid subchain) and resumed in a fresh,
    identically configured engine continues bit for bit."""
    rng = np.random.default_rng(12)
    d, N = 6, 20
    prior = (np.zeros(d), np.eye(d))

    def make():
        if kind in ("mlda_aem", "mlda_aemd", "mlda_aemd_wide", "mlda_dreamz"):
            # error models over three levels with a common output count (dense; diagonal; diagonal with d > 32, where the base
            # subchains run in the fused level kernel) and DREAM(Z) as the base proposal of a hierarchy (host-sequenced path)
            dd = 40 if kind == "mlda_aemd_wide" else d
            e = eng_mod.Engine(N, dd, seed=9, n_levels=3, block_steps=7)
            e.set_prior(np.zeros(dd), np.eye(dd))
            r2 = np.random.default_rng(1)
            base = r2.standard_normal((16, dd)) / 2
            yy = r2.standard_normal(16)
            for k in range(3):
                Ak = base + 0.03 * (2 - k) * r2.standard_normal((16, dd))
                if kind == "mlda_aem" and k < 2:
                    e.set_level(k, Ak, yy, 3, 0.5 * np.eye(16))  # AdaptiveGaussianLogLike below the finest level
                else:
                    e.set_level(k, Ak, yy, 0, 0.5)
            if kind == "mlda_dreamz":
                e.set_proposal_dreamz(12, delta=2, adaptive=True, period=10, capacity=12 + 400)
                e.set_archive(np.random.default_rng(2).standard_normal((N, 12, dd)))
            elif kind == "mlda_aemd_wide":
                e.set_proposal(0, 0.01 * np.eye(dd), scaling=1.0)
            else:
                e.set_proposal(2, 0.05 * np.eye(dd), t0=10, period=10, adaptive=True)
            e.set_subchains([3, 2])
            if kind != "mlda_dreamz":
                e.set_error_model("state-independent" if kind == "mlda_aem" else "state-independent-diagonal")
            e.init(np.full((N, dd), 0.1))
            return e
        if kind in ("da_aem_sd", "da_random"):
            e = eng_mod.Engine(N, d, seed=9, n_levels=2, block_steps=7)
            e.set_prior(*prior)
            r2 = np.random.default_rng(1)
            base = r2.standard_normal((12, d)) / 2
            yy = r2.standard_normal(12)
            if kind == "da_aem_sd":
                e.set_level(0, base + 0.05 * r2.standard_normal((12, d)), yy, 3, 0.5 * np.eye(12))
            else:
                e.set_level(0, base + 0.05 * r2.standard_normal((12, d)), yy, 0, 0.5)
            e.set_level(1, base, yy, 0, 0.5)
            e.set_proposal(2, 0.05 * np.eye(d), t0=10, period=10, adaptive=True)
            if kind == "da_random":
                e.set_subchains([4], randomize=True)
            else:
                e.set_subchains([3])
                e.set_error_model("state-dependent")
            e.init(np.full((N, d), 0.1))
            return e
        if kind == "dream_shared":
            e = eng_mod.Engine(N, d, seed=9, block_steps=7)
            e.set_prior(*prior)
            e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
            e.set_proposal_dreamz(12, delta=2, adaptive=True, period=10, shared=True, sync_every=4, capacity=12 + 80 * N)
            e.set_archive(np.random.default_rng(2).standard_normal((12, d)))
            e.init(np.full((N, d), 0.1))
            return e
        if kind == "mlda":
            e = eng_mod.Engine(N, d, seed=9, n_levels=3, block_steps=7)
            e.set_prior(*prior)
            r2 = np.random.default_rng(1)
            for k, m in enumerate((8, 12, 20)):
                e.set_level(k, r2.standard_normal((m, d)) / 2, r2.standard_normal(m), 0, 0.5)
            e.set_proposal(2, 0.05 * np.eye(d), t0=10, period=10, adaptive=True)
            e.set_subchains([3, 2])
        elif kind == "dreamz":
            e = eng_mod.Engine(N, d, seed=9, block_steps=7)
            e.set_prior(*prior)
            r2 = np.random.default_rng(1)
            e.set_level(0, r2.standard_normal((10, d)) / 2, r2.standard_normal(10), 0, 0.5)
            e.set_proposal_dreamz(12, delta=2, adaptive=True, period=10, capacity=12 + 80)
            e.set_archive(np.random.default_rng(2).standard_normal((N, 12, d)))
        else:
            e = eng_mod.Engine(N, d, seed=9, block_steps=7)
            e.set_prior(*prior)
            r2 = np.random.default_rng(1)
            e.set_level(0, r2.standard_normal((10, d)) / 2, r2.standard_normal(10), 1, 0.3 + r2.random(10))
            e.set_proposal(2, 0.05 * np.eye(d), t0=10, period=10, adaptive=True)
        e.init(np.full((N, d), 0.1))
        return e

    levels = kind.startswith("mlda") or kind.startswith("da_")
    run = (lambda e, n: e.run_levels_host(n)) if levels else (lambda e, n: e.run_host(n))
    flat = (lambda o: [a for lvl in o for a in lvl]) if levels else (lambda o: list(o))
    a = make()
    run(a, 33)
    blob = a.get_state()
    tail_ref = flat(run(a, 27))
    a.close()
    b = make()
    b.set_state(blob)
    tail = flat(run(b, 27))
    for x, y in zip(tail, tail_ref):
        assert np.array_equal(x, y)
    with pytest.raises(Exception):
        b.set_state(blob[:-8])
    b.close()


def test_pinned_host_records_equal_pageable(eng_mod):
    """Records into page-locked host memory take the asynchronous path (second stream, two sets of block buffers); the
    bytes are those of the synchronous pageable path."""
    rng = np.random.default_rng(3)
    d, N, T = 9, 37, 95

    def make():
        e = eng_mod.Engine(N, d, seed=21, block_steps=16)
        e.set_prior(np.zeros(d), np.eye(d))
        r2 = np.random.default_rng(1)
        e.set_level(0, r2.standard_normal((14, d)) / 3, r2.standard_normal(14), 0, 0.3)
        e.set_proposal(2, 0.05 * np.eye(d), t0=20, period=10, adaptive=True)
        e.init(np.full((N, d), 0.1))
        return e

    a = make()
    pa, sa, aa = np.empty((T, N, d)), np.empty((T, N, 3)), np.empty((T, N), dtype=np.uint8)
    a.run(T, pa, sa, aa)
    a.close()
    b = make()
    pb, sb, ab = (eng_mod.pinned_empty((T, N, d)), eng_mod.pinned_empty((T, N, 3)), eng_mod.pinned_empty((T, N), dtype=np.uint8))
    b.run(40, pb[:40], sb[:40], ab[:40])
    b.run(T - 40, pb[40:], sb[40:], ab[40:])
    b.close()
    assert np.array_equal(pa, pb) and np.array_equal(sa, sb) and np.array_equal(aa, ab)


def test_proposal_covariance_set_at_run_time_survives_a_checkpoint(eng_mod):
    """tda_engine_set_proposal_covariance on a two-level GaussianRandomWalk engine that started from the identity (increments =
    the normals themselves, k_rng_direct): the new factor takes effect, and a checkpoint taken afterwards restores it into an
    engine that was again initialised with the identity."""
    d, N = 8, 32
    r2 = np.random.default_rng(3)
    A0, A1 = r2.standard_normal((10, d)) / 2, r2.standard_normal((24, d)) / 2
    y0, y1 = r2.standard_normal(10), r2.standard_normal(24)
    C = np.diag(np.linspace(0.2, 2.0, d))

    def make():
        e = eng_mod.Engine(N, d, seed=21, n_levels=2)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level(0, A0, y0, 0, 0.5)
        e.set_level(1, A1, y1, 0, 0.5)
        e.set_proposal(0, np.eye(d), scaling=0.1)
        e.set_subchains([4])
        e.init(np.zeros((N, d)))
        return e

    flat = lambda outs: [a for lvl in outs for a in lvl]
    plain = make()
    ident_run = flat(plain.run_levels_host(6))
    plain.close()
    a = make()
    a.set_proposal_covariance(C)
    first = flat(a.run_levels_host(6))
    assert not np.array_equal(first[0], ident_run[0])  # the factor is in use
    # steps of level 0 are scaling * L z: per-dimension spread of the first increments follows diag(C)
    blob = a.get_state()
    ref = flat(a.run_levels_host(6))
    a.close()
    b = make()
    b.set_state(blob)
    got = flat(b.run_levels_host(6))
    b.close()
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind", ["mlda", "dreamz", "da_aem"])
def test_pageable_host_records_equal_pinned_in_the_other_drivers(eng_mod, kind):
    """run() / run_levels() into ordinary (pageable) NumPy arrays against page-locked ones for the multilevel, DREAM(Z) and
    error-model drivers (every *_host helper of the wrapper uses page-locked buffers, so the pageable branch needs its own
    case), with the run split at an awkward point."""
    d, N, T = 7, 24, 23

    def make():
        r2 = np.random.default_rng(1)
        if kind == "dreamz":
            e = eng_mod.Engine(N, d, seed=4, block_steps=8)
            e.set_prior(np.zeros(d), np.eye(d))
            e.set_level(0, r2.standard_normal((10, d)) / 2, r2.standard_normal(10), 0, 0.5)
            e.set_proposal_dreamz(12, delta=2, adaptive=True, period=10, capacity=12 + 2 * T)
            e.set_archive(np.random.default_rng(2).standard_normal((N, 12, d)))
        else:
            nl = 3 if kind == "mlda" else 2
            e = eng_mod.Engine(N, d, seed=4, n_levels=nl, block_steps=8)
            e.set_prior(np.zeros(d), np.eye(d))
            base, yy = r2.standard_normal((12, d)) / 2, r2.standard_normal(12)
            for k in range(nl):
                Ak = base + 0.04 * (nl - 1 - k) * r2.standard_normal((12, d))
                if kind == "da_aem" and k == 0:
                    e.set_level(k, Ak, yy, 3, 0.5 * np.eye(12))
                else:
                    e.set_level(k, Ak, yy, 0, 0.5)
            e.set_proposal(2, 0.05 * np.eye(d), t0=10, period=10, adaptive=True)
            e.set_subchains([3, 2] if nl == 3 else [3])
            if kind == "da_aem":
                e.set_error_model("state-independent")
        e.init(np.full((N, d), 0.1))
        return e

    def bufs(e, n, pinned):
        mk = eng_mod.pinned_empty if pinned else (lambda shape, dtype=np.float64: np.empty(shape, dtype=dtype))
        rows = e.rows_per_level(n) if kind != "dreamz" else [n]
        return [(mk((r, N, d)), mk((r, N, 3)), mk((r, N), dtype=np.uint8)) for r in rows]

    got = {}
    for pinned in (True, False):
        e = make()
        parts = []
        for n in (9, T - 9):
            o = bufs(e, n, pinned)
            if kind == "dreamz":
                e.run(n, *o[0])
            else:
                e.run_levels(n, o)
            parts.append(o)
        e.close()
        got[pinned] = [np.concatenate([part[lv][i] for part in parts]) for lv in range(len(parts[0])) for i in range(3)]
    for x, y in zip(got[True], got[False]):
        assert np.array_equal(x, y)
