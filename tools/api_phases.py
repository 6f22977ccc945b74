"""Where does the time of the FIRST and of a later tda.sample() call go?  Phase stamps around every step _sample_device takes
(BASELINE config 2 size: 4096 chains, d = 64, m = 1024, AdaptiveMetropolis; T from argv, default 2000).  One JSON line per call."""
import json
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.stats as st  # noqa: E402

T_NP = time.perf_counter()
import torch  # noqa: E402

T_TORCH = time.perf_counter()
import tinyda_amd as tda  # noqa: E402
from tinyda_amd import _lib  # noqa: E402
from tinyda_amd.engine import Engine, pinned_empty  # noqa: E402

T_PKG = time.perf_counter()


def stamp(rec, name, t):
    now = time.perf_counter()
    rec[name] = round(now - t, 4)
    return now


def one_call(T, N=4096, d=64, m=1024, records="pinned"):
    rec = {"T": T, "N": N, "records": records}
    rng = np.random.default_rng(1)
    A = rng.standard_normal((m, d)) / 8
    y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
    t = time.perf_counter()
    t_all = t
    _lib.load()
    t = stamp(rec, "lib_load", t)
    torch.cuda.init()
    torch.empty(1, device="cuda:0")
    torch.cuda.synchronize()
    t = stamp(rec, "torch_cuda_init", t)
    e = Engine(N, d, seed=1)
    t = stamp(rec, "engine_create", t)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    t = stamp(rec, "set_problem", t)
    e.init(None)
    e.sync()
    t = stamp(rec, "init", t)
    if records == "pinned":
        params, stat, acc = pinned_empty((T + 1, N, d)), pinned_empty((T + 1, N, 3)), pinned_empty((T + 1, N), dtype=np.uint8)
        p1, s1, a1 = params[1:], stat[1:], acc[1:]
    else:
        params = torch.empty((T + 1, N, d), dtype=torch.float64, device="cuda:0")
        stat = torch.empty((T + 1, N, 3), dtype=torch.float64, device="cuda:0")
        acc = torch.empty((T + 1, N), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        p1, s1, a1 = params[1:], stat[1:], acc[1:]
    t = stamp(rec, "alloc_records", t)
    e.run(T, p1, s1, a1)
    t = stamp(rec, "run", t)
    e.proposal_state(want_am=True)
    t = stamp(rec, "proposal_state", t)
    e.close()
    t = stamp(rec, "close", t)
    rec["total"] = round(t - t_all, 4)
    rec["evals_per_s_run"] = N * T / rec["run"]
    return rec


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    print(json.dumps({"import_numpy_scipy": round(T_NP - T0, 3), "import_torch": round(T_TORCH - T_NP, 3), "import_pkg": round(T_PKG - T_TORCH, 3)}))
    for records in ("pinned", "pinned", "device", "device"):
        print(json.dumps(one_call(T, records=records)), flush=True)
