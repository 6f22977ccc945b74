"""Step-kernel time per evaluation against the number of observations for the 4-wave and 8-wave tiles (single level,
pCN, 4096 chains, d = 64): what an 8-wave tile could give the multi-level kernel, which still runs the 4-wave one.
Run once per setting: TINYDA_STEPS_WAVES=4|8 python tools/waves_vs_m.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine

N, d, T = 4096, 64, 400
for m in (128, 256, 512, 1024, 2048):
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, d)) / 8
    y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
    e = Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(1, None, scaling=0.02)
    e.init(None)
    params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    stats = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    e.run(100, params[:100], stats[:100], acc[:100])
    e.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run(T, params, stats, acc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = e.profile()
    ev = N * T
    fl = 2 * m * d + 3 * m + 2 * d
    print("waves=%s m=%4d: %.3e evals/s; steps kernel %.3f ns/eval = %.1f TFLOP/s; propose %.3f ns/eval" % (
        os.environ.get("TINYDA_STEPS_WAVES", "8"), m, ev / dt, p["ms_steps"] * 1e6 / ev, ev * fl / (p["ms_steps"] * 1e-3) * 1e-12,
        p["ms_propose"] * 1e6 / ev))
    e.close()
