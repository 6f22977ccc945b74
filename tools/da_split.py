"""Cost of a coarse step and of a level action in the Delayed-Acceptance / MLDA kernels: the C3 / C5 shapes at two subchain
lengths (same number of finest iterations), so that the difference is pure coarse steps.  python tools/da_split.py [c3|c5]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine
from bench_configs import levels


def steps_ms(ms, sl, prop, n_fine, N=4096, d=64):
    lv = levels(ms)
    e = Engine(N, d, seed=9, n_levels=len(ms))
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(**prop)
    e.set_subchains(sl)
    e.init(None)
    e.run_levels(max(1, n_fine // 10), None)
    e.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run_levels(n_fine, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = e.profile()
    rows = e.rows_per_level(n_fine)
    e.close()
    return p["ms_steps"], rows, dt * 1e3


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "c3"
    if what == "c3":
        cases = [((256, 2048), [10]), ((256, 2048), [20]), ((256, 2048), [40]), ((128, 2048), [10]), ((128, 2048), [20])]
        prop, n_fine = dict(kind=1, scaling=0.02), 200
    else:
        cases = [((128, 512, 2048), [5, 3]), ((128, 512, 2048), [10, 3]), ((128, 512, 2048), [20, 3]), ((128, 512, 2048), [5, 6])]
        prop, n_fine = dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100), 60
    out = []
    for ms, sl in cases:
        t, rows, wall = steps_ms(ms, sl, prop, n_fine)
        out.append(dict(observations=ms, subchains=sl, rows=rows, ms_steps=t, wall_ms=wall, us_per_coarse_step=1e3 * t / rows[0]))
        print(json.dumps(out[-1]), flush=True)
