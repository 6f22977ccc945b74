"""Experiment: G engines with N/G chains each on separate HIP streams, run() enqueued asynchronously, vs one engine."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine

rng = np.random.default_rng(1)
d, m, N, T = 64, 1024, 4096, 2000
A = rng.standard_normal((m, d)) / 8
th = rng.standard_normal(d)
y = A @ th + 0.1 * rng.standard_normal(m)

def make(n, off):
    e = Engine(n, d, seed=1, chain_offset=off)
    e.set_prior(np.zeros(d), np.eye(d)); e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.init(None)
    bufs = (torch.empty((T, n, d), dtype=torch.float64, device="cuda"), torch.empty((T, n, 3), dtype=torch.float64, device="cuda"),
            torch.empty((T, n), dtype=torch.uint8, device="cuda"))
    return e, bufs

for G in (1, 2, 4, 1, 2, 4):
    engs = [make(N // G, g * (N // G)) for g in range(G)]
    for e, b in engs:
        e.run(200, b[0][:200], b[1][:200], b[2][:200])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e, b in engs:
        e.run(T, *b, sync=False)
    for e, _ in engs:
        e.sync()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("groups=%d  %.3e evals/s" % (G, N * T / dt))
    for e, _ in engs:
        e.close()
