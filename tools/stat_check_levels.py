"""Do Delayed Acceptance and MLDA leave the FINEST posterior invariant?  Linear-Gaussian hierarchies (conjugate finest posterior),
4096 chains started from exact draws of it, CrankNicolson base proposals (nothing adapts): the pooled mean and variances of the
finest chain's recorded states must stay at the closed form.  One JSON line per hierarchy."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tinyda_amd import _lib
from tinyda_amd.engine import Engine

_lib.load()
D, N, SIGMA = 64, 4096, 0.1


def check(name, ms, sl, n_fine, beta):
    rng = np.random.default_rng(7)
    truth = rng.standard_normal(D)
    lv = []
    for m in ms:  # every level its own observation operator and noise of the same truth (the recipe of tools/bench_configs.py)
        A = rng.standard_normal((m, D)) / 8
        lv.append((A, A @ truth + SIGMA * rng.standard_normal(m)))
    Af, yf = lv[-1]
    cov_post = np.linalg.inv(Af.T @ Af / SIGMA ** 2 + np.eye(D))
    mean_post = cov_post @ (Af.T @ yf / SIGMA ** 2)
    e = Engine(N, D, seed=31, n_levels=len(ms))
    e.set_prior(np.zeros(D), np.eye(D))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, SIGMA ** 2)
    e.set_proposal(1, None, scaling=beta)
    e.set_subchains(sl)
    e.init(mean_post + (np.linalg.cholesky(cov_post) @ rng.standard_normal((D, N))).T)
    outs = e.run_levels_host(n_fine)
    e.close()
    P, S, A_ = outs[-1]
    half = P[n_fine // 2:]
    flat = half.reshape(-1, D)
    r = flat.var(axis=0) / np.diag(cov_post)
    # the second half's chain means are (nearly) independent across chains: z of the pooled mean from the spread of chain means
    cm = half.mean(axis=0)
    z = (cm.mean(axis=0) - mean_post) / (cm.std(axis=0, ddof=1) / np.sqrt(N))
    print(json.dumps(dict(check=name, fine_iterations=n_fine, acceptance_per_level=[float(o[2].mean()) for o in outs],
                          var_ratio_min=float(r.min()), var_ratio_max=float(r.max()), var_ratio_mean=float(r.mean()),
                          max_abs_z_of_mean=float(np.abs(z).max()), rms_z_of_mean=float(np.sqrt((z ** 2).mean())))))


check("Delayed Acceptance, pCN(0.003), 256 / 2048 observations, subchain 10 (k_da_steps)", (256, 2048), [10], 200, 0.003)
check("MLDA, pCN(0.003), 128 / 512 / 2048 observations, subchains [5, 3] (k_da_steps<.., 3>)", (128, 512, 2048), [5, 3], 120, 0.003)


def check_dream(name, peer):
    """DREAM with one shared archive on a conjugate target (d = 16, 64 observations): started from exact posterior draws, the archive
    seeded with exact draws too; pooled variance of the second half of the run"""
    import torch

    from tinyda_amd import distributed as tdist

    d, m, n, T, M0, K = 16, 64, 4096, 600, 2048, 16
    rng = np.random.default_rng(9)
    A = rng.standard_normal((m, d)) / 4
    truth = rng.standard_normal(d)
    y = A @ truth + SIGMA * rng.standard_normal(m)
    cov_post = np.linalg.inv(A.T @ A / SIGMA ** 2 + np.eye(d))
    mean_post = cov_post @ (A.T @ y / SIGMA ** 2)
    Lp = np.linalg.cholesky(cov_post)
    e = Engine(n, d, seed=41)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, SIGMA ** 2)
    e.set_proposal_dreamz(M0, delta=1, nCR=3, adaptive=False, shared=True, sync_every=K, capacity=M0 + T * n)
    e.set_archive(mean_post + (Lp @ rng.standard_normal((d, M0))).T)
    e.init(mean_post + (Lp @ rng.standard_normal((d, n))).T)
    p = torch.empty((T, n, d), dtype=torch.float64, device="cuda")
    a = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    if peer:
        tdist.setup_peer_archive(e)
        tdist.run_peer_dream(e, T, K, p, None, a)
    else:
        tdist.run_shared_dream(e, T, K, p, None, a)
    e.sync()
    e.close()
    half = p[T // 2:]
    flat = half.reshape(-1, d)
    r = (flat.var(dim=0).cpu().numpy()) / np.diag(cov_post)
    cm = half.mean(dim=0).cpu().numpy()
    z = (cm.mean(axis=0) - mean_post) / (cm.std(axis=0, ddof=1) / np.sqrt(n))
    print(json.dumps(dict(check=name, steps=T, acceptance=float(a.float().mean()), var_ratio_min=float(r.min()), var_ratio_max=float(r.max()),
                          var_ratio_mean=float(r.mean()), max_abs_z_of_mean=float(np.abs(z).max()), rms_z_of_mean=float(np.sqrt((z ** 2).mean())))))


check_dream("DREAM, shared replicated archive, conjugate target d = 16", False)
check_dream("DREAM, shared DISTRIBUTED archive (one rank), conjugate target d = 16", True)
