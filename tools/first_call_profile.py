"""cProfile of the FIRST tda.sample() call of a fresh process (torch not imported by the script): where do its seconds go?"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.stats as st

import tinyda_amd as tda

d, m, N, T = 64, 1024, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(1)
A = rng.standard_normal((m, d)) / 8
y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(m)), tda.LinearModel(A))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
pr.disable()
print("first sample(): %.2f s; torch imported: %s" % (time.perf_counter() - t0, "torch" in sys.modules))
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
t0 = time.perf_counter()
res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
print("second sample(): %.3f s" % (time.perf_counter() - t0))
