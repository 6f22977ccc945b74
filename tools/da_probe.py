"""Where does the time of the two-level DA kernel go?  Same 2000 coarse steps with different subchain lengths (fewer fine evaluations)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinyda_amd.engine import Engine
from tools.bench_configs import levels

def run(L, n_fine, ms=(256, 2048), N=4096, d=64):
    lv = levels(ms)
    e = Engine(N, d, seed=9, n_levels=2)
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(kind=1, scaling=0.02)
    e.set_subchains([L])
    e.init(None)
    e.run_levels(max(1, n_fine // 10), None)
    e.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run_levels(n_fine, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = e.profile()
    print("m=%s L=%d n_fine=%d: steps kernel %.2f ms (%d launches), propose %.2f ms, wall %.2f ms -> %.2f us per coarse step" % (
        ms, L, n_fine, p["ms_steps"], p["n_launch_steps"], p["ms_propose"], dt * 1e3, p["ms_steps"] * 1e3 / (L * n_fine)), flush=True)
    e.close()

for L, nf in ((10, 200), (100, 20), (1000, 2)):
    run(L, nf)
run(10, 200, ms=(128, 2048))
run(10, 200, ms=(256, 256))
