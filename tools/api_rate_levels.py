"""What a user of tda.sample() gets on the multi-level BASELINE configurations (C3: Delayed Acceptance, C5-literal: 3-level MLDA),
end to end: engine set-up, the run, records in HBM, the result dict.  One JSON line per configuration."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.stats as st  # noqa: E402

import tinyda_amd as tda  # noqa: E402

d, N = 64, 4096


def posts(ms, seed=2, sigma=0.1):
    rng = np.random.default_rng(seed)
    truth = rng.standard_normal(d)
    prior = st.multivariate_normal(np.zeros(d), np.eye(d))
    out = []
    for m in ms:
        A = rng.standard_normal((m, d)) / 8
        out.append(tda.Posterior(prior, tda.GaussianLogLike(A @ truth + sigma * rng.standard_normal(m), sigma ** 2 * np.eye(m)), tda.LinearModel(A)))
    return out


def rate(name, ms, proposal, n_fine, calls=4, **kw):
    ps = posts(ms)
    times = []
    for _ in range(calls):
        t0 = time.perf_counter()
        res = tda.sample(ps, proposal(), n_fine, n_chains=N, seed=1, **kw)
        times.append(time.perf_counter() - t0)
    sl = kw.get("subchain_length")
    sl = [sl] * (len(ms) - 1) if np.isscalar(sl) else list(sl)
    coarse = n_fine * int(np.prod(sl))
    steady = min(times[1:])
    return dict(name=name, backend=res.get("backend"), sample_seconds=[round(t, 4) for t in times], steady_seconds=round(steady, 4),
                coarse_evals_per_s_end_to_end=N * coarse / steady)


if __name__ == "__main__":
    print(json.dumps(rate("C3: DA, pCN(0.02), 256 / 2048 observations, subchain 10, 200 finest iterations", (256, 2048),
                          lambda: tda.CrankNicolson(scaling=0.02), 200, subchain_length=10)), flush=True)
    print(json.dumps(rate("C5-literal: MLDA, AM, 128 / 512 / 2048 observations, subchains [5, 3], 60 finest iterations", (128, 512, 2048),
                          lambda: tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), 60, subchain_length=[5, 3])), flush=True)
