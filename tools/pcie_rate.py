"""C2a with HOST record buffers (NumPy): every block's records cross PCIe (pageable memory). Reported in DESIGN.md only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyda_amd.engine import Engine
d, m, N, T = 64, 1024, 4096, 600
rng = np.random.default_rng(1)
A = rng.standard_normal((m, d)) / 8
y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
e = Engine(N, d, seed=1)
e.set_prior(np.zeros(d), np.eye(d)); e.set_level(0, A, y, 0, 0.01)
e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
e.init(None)
P, S, Acc = np.empty((T, N, d)), np.empty((T, N, 3)), np.empty((T, N), dtype=np.uint8)
e.run(100, P[:100], S[:100], Acc[:100])
t0 = time.perf_counter(); e.run(T, P, S, Acc); dt = time.perf_counter() - t0
print("host (pageable NumPy) record buffers: %.3e evals/s, %.2f GB/s of records over PCIe" % (N * T / dt, N * T * 537 / dt / 1e9))
t0 = time.perf_counter(); e.run(T, None, S, None); dt = time.perf_counter() - t0
print("host stats only (24 B/eval)          : %.3e evals/s" % (N * T / dt))
from tinyda_amd.engine import pinned_empty
P, S, Acc = pinned_empty((T, N, d)), pinned_empty((T, N, 3)), pinned_empty((T, N), dtype=np.uint8)
e.run(100, P[:100], S[:100], Acc[:100])
t0 = time.perf_counter(); e.run(T, P, S, Acc); dt = time.perf_counter() - t0
print("host (pinned) record buffers, copies on a second stream: %.3e evals/s, %.2f GB/s of records over PCIe" % (N * T / dt, N * T * 537 / dt / 1e9))
e.close()
