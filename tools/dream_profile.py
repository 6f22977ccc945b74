"""Where does a DREAM (shared archive) block go at the C4 shape?  Wall time of run_shared_dream against the engine's own
per-kernel HIP-event timings, for several exchange intervals K."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tinyda_amd import distributed as tdist  # noqa: E402
from tinyda_amd.engine import Engine  # noqa: E402

N, d, M0 = 8192, 32, 320
T = int(sys.argv[1]) if len(sys.argv) > 1 else 384
for K in (16, 64, 128):
    e = Engine(N, d, seed=4)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
    e.set_proposal_dreamz(M0, delta=1, nCR=3, adaptive=True, period=128, shared=True, sync_every=K, capacity=M0 + (T + 3 * K + 16) * N)
    e.set_archive(None)
    e.init(None)
    params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    stats = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    tdist.run_shared_dream(e, 2 * K, K, params, stats, acc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tdist.run_shared_dream(e, T, K, params, stats, acc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # one more block with the engine's event timers on
    e.set_profiling(True)
    e.run(K, params[:K], stats[:K], acc[:K], sync=True)
    prof = e.profile()
    print(json.dumps(dict(K=K, evals_per_s=N * T / dt, us_per_step=dt / T * 1e6, acceptance=float(acc.float().mean().item()),
                          one_block_kernel_ms=dict(draw=prof["ms_propose"], steps=prof["ms_steps"], adapt=prof["ms_adapt"]),
                          one_block_kernel_us_per_step=(prof["ms_propose"] + prof["ms_steps"] + prof["ms_adapt"]) * 1e3 / K)))
    e.close()
