"""cProfile of a steady-state tda.sample() call at BASELINE config 2 size (the third call of the process)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.stats as st

import tinyda_amd as tda

d, m, N, T = 64, 1024, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(1)
A = rng.standard_normal((m, d)) / 8
y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(m)), tda.LinearModel(A))
for rep in range(3):
    t0 = time.perf_counter()
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
    t1 = time.perf_counter()
    print("sample(): %.4f s (%.2e evals/s end to end)" % (t1 - t0, N * T / (t1 - t0)))
pr = cProfile.Profile()
pr.enable()
res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
t0 = time.perf_counter()
s = tda.get_samples(res, burnin=T // 2)
print("get_samples: %.3f s = %.2f GB/s" % (time.perf_counter() - t0, N * s["chain_0"].nbytes / (time.perf_counter() - t0) / 1e9))
