"""Do two half-size engines on separate streams (driven by two host threads) beat one full-size engine?  The step kernel is
bound by the matrix cores, the adaptation kernels by the VALU, so phase-shifted groups could overlap on the same CUs."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tinyda_amd.engine import Engine  # noqa: E402

D, M = 64, 1024
rng = np.random.default_rng(1)
A = rng.standard_normal((M, D)) / 8
y = A @ rng.standard_normal(D) + 0.1 * rng.standard_normal(M)


def make(N, offset):
    e = Engine(N, D, seed=2026, chain_offset=offset)
    e.set_prior(np.zeros(D), np.eye(D))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(D), t0=100, period=100)
    e.init(None)
    bufs = (torch.empty((2000, N, D), dtype=torch.float64, device="cuda"), torch.empty((2000, N, 3), dtype=torch.float64, device="cuda"),
            torch.empty((2000, N), dtype=torch.uint8, device="cuda"))
    e.run(200, bufs[0][:200], bufs[1][:200], bufs[2][:200], sync=True)
    return e, bufs


def timed(engines, K, stagger=0):
    def work(e, b, delay):
        if delay:
            e.run(delay, b[0][:delay], b[1][:delay], b[2][:delay])
        e.run(K, b[0][:K], b[1][:K], b[2][:K], sync=True)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(e, b, 0)) for i, (e, b) in enumerate(engines)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


K = 2000
one = [make(4096, 0)]
dt = timed(one, K)
print("1 x 4096 chains: %.3e evals/s" % (4096 * K / dt))
one[0][0].close()
del one
for groups in (2, 4):
    n = 4096 // groups
    eng = [make(n, i * n) for i in range(groups)]
    dt = timed(eng, K)
    print("%d x %d chains: %.3e evals/s" % (groups, n, 4096 * K / dt))
    for e, _ in eng:
        e.close()
    del eng
eng = [make(4096, i * 4096) for i in range(2)]
dt = timed(eng, K)
print("2 x 4096 chains: %.3e evals/s" % (2 * 4096 * K / dt))
