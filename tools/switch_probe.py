"""One seeded run of a configuration whose kernel choice an environment switch changes; records to an .npz.  The switches are
read once per process, so A/B comparisons start this script twice (tests/test_gpu_switches.py).
    python tools/switch_probe.py {mlda3|mlda3_short|da2|aemd|aemd_lean|aem_dense|aem_dense_da_pcn|aem_dense_m200|aem_dense_chunks|dream|am}[_ragged] out.npz     (_ragged: a chain count that is not a multiple of the 16-chain tile)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tinyda_amd import _lib, engine


def hierarchy(ms, sl, kind, n_fine, N=256, d=64, error_model=None):
    rng = np.random.default_rng(11)
    truth = rng.standard_normal(d)
    e = engine.Engine(N, d, seed=9, n_levels=len(ms))
    e.set_prior(np.zeros(d), np.eye(d))
    base = rng.standard_normal((ms[-1], d)) / 8
    for k, m in enumerate(ms):
        A = base[:m] + (0.02 * (len(ms) - 1 - k) * rng.standard_normal((m, d)) / 8 if error_model else 0.0) if error_model else rng.standard_normal((m, d)) / 8
        if error_model in ("state-independent", "state-dependent") and k < len(ms) - 1:  # dense error model: AdaptiveGaussianLogLike below the finest level
            e.set_level(k, A, A @ truth + 0.1 * rng.standard_normal(m), 3, 0.01 * np.eye(m))
        else:
            e.set_level(k, A, A @ truth + 0.1 * rng.standard_normal(m), 0, 0.01)
    if kind == "am":
        e.set_proposal(2, 1e-4 * np.eye(d), t0=20, period=20)
    else:
        e.set_proposal(1, None, scaling=0.02)
    e.set_subchains(sl)
    if error_model:
        e.set_error_model(error_model)
    e.init(truth + 0.05 * rng.standard_normal((N, d)))
    outs = e.run_levels_host(n_fine)
    e.close()
    return {"%s%d" % (n, k): o[i] for k, o in enumerate(outs) for i, n in enumerate(("params", "stats", "acc"))}


def dream(N=512, d=32, T=70, M0=64, K=16):
    e = engine.Engine(N, d, seed=8)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
    e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=True, period=32, shared=True, sync_every=K, capacity=M0 + T * N)
    e.set_archive(None)
    e.init(None)
    P, S, A = e.run_host(T)
    st = e.dreamz_state()
    e.close()
    return dict(params0=P, stats0=S, acc0=A, pCR=st["pCR"])


def single_am(N=96, d=64, m=200, T=330):
    """single-level AdaptiveMetropolis over several covariance swaps (k_chol_apply against k_chol + k_apply)"""
    rng = np.random.default_rng(12)
    A = rng.standard_normal((m, d)) / 8
    truth = rng.standard_normal(d)
    e = engine.Engine(N, d, seed=10)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, A @ truth + 0.1 * rng.standard_normal(m), 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=60, period=60, adaptive=True)
    e.init(truth + 0.05 * rng.standard_normal((N, d)))
    P, S, Acc = e.run_host(T)
    st = e.proposal_state(want_am=True)
    e.close()
    return dict(params0=P, stats0=S, acc0=Acc, C=st["C"], sigma=st["am_sigma"], scaling=st["scaling"])


if __name__ == "__main__":
    _lib.load()
    what, out = sys.argv[1], sys.argv[2]
    ragged = what.endswith("_ragged")
    what = what[:-7] if ragged else what
    cut = 7 if ragged else 0  # the last tile holds 9 chains
    if what == "mlda3":
        res = hierarchy((128, 256, 512), [5, 3], "am", 6, N=256 - cut)
    elif what == "mlda3_short":  # run() calls that end inside an adaptation period (20 base steps): 15 base steps per finest iteration
        res = hierarchy((128, 256, 512), [5, 3], "am", 5, N=64)
    elif what == "da2":
        res = hierarchy((256, 1024), [10], "pcn", 8, N=256 - cut)
    elif what == "aemd":
        res = hierarchy((200, 200, 200), [5, 3], "am", 8, N=96 - cut, error_model="state-independent-diagonal")
    elif what == "aemd_lean":  # at most 128 outputs: the base subchains are eligible for k_da_steps
        res = hierarchy((128, 128, 128), [5, 3], "am", 8, N=96 - cut, error_model="state-independent-diagonal")
    elif what == "aem_dense":  # dense error model, three levels: base subchains on k_aem_base_steps / k_ml_steps
        res = hierarchy((100, 100, 100), [5, 3], "am", 8, N=96 - cut, error_model="state-independent")
    elif what == "aem_dense_da_pcn":  # two levels, pCN (keep < 1 in the linear update), state-dependent model
        res = hierarchy((40, 40), [4], "pcn", 12, N=64, error_model="state-dependent")
    elif what == "aem_dense_m200":  # 129 .. 256 outputs (round 5): k_aem_refresh_big, k_aem_action<256>, k_aem_base_steps<16> / aem_quad_factor_inplace<16>
        res = hierarchy((200, 200, 200), [5, 3], "am", 6, N=48 - cut, error_model="state-independent")
    elif what == "aem_dense_chunks":  # a 20-step base subchain: 22 vectors, two chunks of 16 columns in k_aem_base_steps
        res = hierarchy((72, 72), [20], "pcn", 4, N=32, error_model="state-independent")
    elif what == "aem_dense_long":  # a base subchain whose product vectors do not fit 64 KB of LDS (128 outputs, 70 steps): level kernel
        res = hierarchy((128, 128), [70], "pcn", 2, N=32, error_model="state-independent")
    elif what == "am":
        res = single_am(N=96 - cut)
    elif what in ("am_d40", "am_d33"):  # fewer than 64 parameters on the 64-parameter instances: padded rows / columns of Sigma
        res = single_am(N=96 - cut, d=int(what[4:]), m=120, T=250)
    else:
        res = dream(N=512 - cut)
    np.savez(out, **res)
