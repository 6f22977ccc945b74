"""What a user of tda.sample() gets at BASELINE config 2 (4096 chains, d = 64, m = 1024, AdaptiveMetropolis, T iterations):
the first call of a fresh process (torch NOT imported by the script: the package imports it before the HIP runtime starts), later
calls, get_samples of everything, one lazy chain, the lazy proposal state.  One JSON object on stdout (profiles/r03_api.json)."""
import json
import os
import sys
import time

T_START = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.stats as st  # noqa: E402

import tinyda_amd as tda  # noqa: E402

d, m, N = 64, 1024, 4096
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(1)
A = rng.standard_normal((m, d)) / 8
y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
post = tda.Posterior(st.multivariate_normal(np.zeros(d), np.eye(d)), tda.GaussianLogLike(y, 0.01 * np.eye(m)), tda.LinearModel(A))
out = {"workload": "BASELINE configs[1]: %d chains, d=%d, m=%d, AdaptiveMetropolis(t0=100, period=100), %d iterations" % (N, d, m, T),
       "import_seconds": round(time.perf_counter() - T_START, 3), "torch_imported_by_script": "torch" in sys.modules}
times = []
res = None
for c in range(calls):
    t0 = time.perf_counter()
    res = tda.sample(post, tda.AdaptiveMetropolis(1e-4 * np.eye(d), t0=100, period=100), T, n_chains=N, seed=1)
    times.append(time.perf_counter() - t0)
out["sample_seconds"] = [round(t, 4) for t in times]
steady = min(times[1:]) if calls > 1 else times[0]
out["first_call_seconds"] = round(times[0], 4)
out["first_call_overhead_seconds"] = round(times[0] - steady, 4)
out["steady_seconds"] = round(steady, 4)
out["evals_per_s_end_to_end"] = N * T / steady
out["evals_per_s_median_call"] = N * T / float(np.median(times[1:])) if calls > 1 else None
t0 = time.perf_counter()
one = res["chain_17"].parameters
out["one_chain_fetch_seconds"] = round(time.perf_counter() - t0, 4)
t0 = time.perf_counter()
s = tda.get_samples(res, burnin=T // 2)
dt = time.perf_counter() - t0
out["get_samples_all_chains_seconds"] = round(dt, 4)
out["get_samples_GB_per_s"] = round(N * s["chain_0"].nbytes / dt / 1e9, 2)
assert np.array_equal(s["chain_17"], one[T // 2:])
t0 = time.perf_counter()
ps = res["proposal_state"]["C"]
out["lazy_proposal_state_seconds"] = round(time.perf_counter() - t0, 4)
out["proposal_state_C_shape"] = list(ps.shape)
out["device_memory_GB_held_by_result"] = round(sum(getattr(res["chain_0"]._records, f).numel() * getattr(res["chain_0"]._records, f).element_size()
                                                   for f in ("parameters", "stats", "accepted")) / 1e9, 3)
out["torch_pinned_or_host_record_bytes"] = 0
print(json.dumps(out))
