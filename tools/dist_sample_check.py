"""Rehearsal of sample(distributed=True) on a one-GPU box: TINYDA_BENCH_ONE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_sample_check.py"""
import sys, numpy as np, scipy.stats as st
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyda_amd as tda
rng = np.random.default_rng(0)
A = rng.standard_normal((12, 6)) / 2; y = rng.standard_normal(12)
post = tda.Posterior(st.multivariate_normal(np.zeros(6), np.eye(6)), tda.GaussianLogLike(y, 0.25 * np.eye(12)), tda.LinearModel(A))
res = tda.sample(post, tda.DREAM(24, adaptive=True, period=20), 40, n_chains=32, seed=5, distributed=True)
print("DIST_OK", res["n_chains"], res["chain_offset"], len(res["chain_0"]), res["proposal_state"]["archive_rows"])
res = tda.sample(post, tda.AdaptiveMetropolis(0.05 * np.eye(6), t0=10, period=10), 40, n_chains=32, seed=5, distributed=True)
print("DIST_OK", res["n_chains"], res["chain_offset"], len(res["chain_%d" % (res["n_chains"] - 1)]))
res = tda.sample(post, tda.DREAM(24, adaptive=True, period=20), 40, n_chains=32, seed=5, distributed=True, overlap_archive_exchange=True)
print("DIST_OVERLAP_OK", res["n_chains"], res["chain_offset"], len(res["chain_0"]), res["proposal_state"]["archive_rows"])
for overlap in (False, True):
    res = tda.sample(post, tda.DREAM(24, adaptive=True, period=20), 40, n_chains=32, seed=5, distributed=True, shared_archive="distributed",
                     overlap_archive_exchange=overlap)
    print("DIST_PEER_OK", overlap, res["n_chains"], res["chain_offset"], len(res["chain_0"]), res["proposal_state"]["archive_rows"])
