"""Throughput of the batched host-callback path (tda.BatchedModel) at the C2a shape: the model is a NumPy matmul on the
host, everything else of the step runs on the device.  Prints evals/s and where a step's time goes."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

from tinyda_amd.engine import Engine

N, d, m = 4096, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(1)
A = rng.standard_normal((m, d)) / 8
truth = rng.standard_normal(d)
y = A @ truth + 0.1 * rng.standard_normal(m)
t_model = [0.0]


def fn(thetas, out):
    t0 = time.perf_counter()
    np.matmul(thetas, A.T, out=out)
    t_model[0] += time.perf_counter() - t0


e = Engine(N, d, seed=1)
e.set_prior(np.zeros(d), np.eye(d))
e.set_level_callback(0, fn, y, 0, [0.01], inplace=True)
e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
e.init(None)
e.run(20)
e.sync()
t_model[0] = 0.0
t0 = time.perf_counter()
e.run(T)
e.sync()
dt = time.perf_counter() - t0
print("callback path: N=%d d=%d m=%d  %d steps in %.3f s = %.3e evals/s; %.2f ms/step of which model %.2f ms"
      % (N, d, m, T, dt, N * T / dt, 1e3 * dt / T, 1e3 * t_model[0] / T))
e.close()
