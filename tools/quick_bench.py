"""Ad-hoc timing of BASELINE config 2a on one GPU with per-kernel HIP-event breakdown (not the bench contract)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine

def problem(d=64, m=1024, seed=1, sigma=0.1):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d)) / 8
    th = rng.standard_normal(d)
    y = A @ th + sigma * rng.standard_normal(m)
    return A, th, y

N, d, m = 4096, 64, 1024
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
A, th, y = problem(d, m)
e = Engine(N, d, seed=1)
e.set_prior(np.zeros(d), np.eye(d))
if len(sys.argv) > 2 and sys.argv[2] == "dense":  # C2b: dense noise covariance (SURVEY 8d variant 2b)
    rng = np.random.default_rng(1)
    Lc = 0.1 * np.eye(m) + 0.01 * np.tril(rng.standard_normal((m, m)))
    e.set_level(0, A, y, 2, Lc @ Lc.T)
else:
    e.set_level(0, A, y, 0, 0.01)
e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100, block_moments="block" in sys.argv)
e.init(None)
params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
stats = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
e.run(200, params[:200], stats[:200], acc[:200])  # warm-up
for prof in (0, 1):
    e.set_profiling(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run(T, params, stats, acc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("profiling=%d  %d steps x %d chains: %.3f s  -> %.3e evals/s  (acc rate %.3f)" % (prof, T, N, dt, N * T / dt, acc.float().mean().item()))
    if prof:
        p = e.profile(); print(p)
        ev = N * T
        print("steps kernel: %.1f ns/eval -> %.2f TFLOP/s fp64 (2md+3m+3d flops/eval)" % (p["ms_steps"] * 1e6 / ev, ev * (2*m*d + 3*m + 3*d) / (p["ms_steps"] * 1e-3) * 1e-12))
        print("propose: %.1f ns/eval   adapt: %.1f ns/eval" % (p["ms_propose"] * 1e6 / ev, p["ms_adapt"] * 1e6 / ev))
