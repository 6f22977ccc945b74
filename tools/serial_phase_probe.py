"""How much of a k_mh_steps step is not MFMA work: the same 4096-chain AM run with m = 16 .. 1024 observations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine
d, N, T = 64, 4096, 1000
for m in (16, 128, 256, 512, 1024, 2048):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((m, d)) / 8
    y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
    e = Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d)); e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.init(None)
    p = torch.empty((T, N, d), dtype=torch.float64, device="cuda"); s = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    a = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    e.run(200, p[:200], s[:200], a[:200])
    e.set_profiling(True)
    e.run(T, p, s, a)
    pr = e.profile()
    us_step = pr["ms_steps"] * 1e3 / T
    mfma_us = (m / 16) * 16 / 4 * 64 / 2.27e3 / 2  # blocks * ksteps / 4 SIMDs... per CU: (m/16*16 MFMA per tile)/4 SIMD * 64 cyc
    print("m=%5d  steps kernel %.2f us/step  (%.2f ns/eval)   ideal MFMA time %.2f us" % (m, us_step, us_step * 1e3 / N, (m / 16) * 16 * 64 / 4 / 2.27e3))
    e.close()
