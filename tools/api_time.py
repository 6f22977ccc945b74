import time, numpy as np, scipy.stats as st, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyda_amd as tda
d, m, N, T = 64, 1024, 4096, 1000
rng = np.random.default_rng(1)
A = rng.standard_normal((m, d)) / 8
y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(m)), tda.LinearModel(A))
for rep in range(2):
    t0 = time.perf_counter()
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
    t1 = time.perf_counter()
    s = tda.get_samples(res, burnin=T // 2)
    t2 = time.perf_counter()
    print("sample(): %.2f s (%.2e evals/s end to end), get_samples: %.2f s, chain_0 %s" % (t1 - t0, N * T / (t1 - t0), t2 - t1, s["chain_0"].shape))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
