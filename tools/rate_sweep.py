"""Throughput of configurations OUTSIDE the BASELINE set, one line each: a screen for instances that run far below their neighbours
(how the two-waves-per-SIMD DREAM(Z) tile kernel at 64 parameters was found).  4096 chains; coarsest-level evaluations per second
with the kernel-time buckets of a profiled repetition.   python tools/rate_sweep.py [filter]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench_configs as bc

N = 4096


def single(name, d, m, prop, T=300, noise=0, prior="identity", dreamz=None):
    torch = bc._torch()
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, d)) / 8
    truth = rng.standard_normal(d)
    y = A @ truth + 0.1 * rng.standard_normal(m)
    e = bc._engine()(N, d, seed=3)
    if prior == "dense":
        B = rng.standard_normal((d, d))
        e.set_prior(np.zeros(d), np.eye(d) + 0.3 * B @ B.T / d)
    else:
        e.set_prior(np.zeros(d), np.eye(d))
    if noise == 1:
        e.set_level(0, A, y, 1, 0.01 * (1 + rng.random(m)))
    elif noise == 2:
        e.set_level(0, A, y, 2, 0.01 * (np.eye(m) + 0.1 * np.ones((m, m)) / m))
    else:
        e.set_level(0, A, y, 0, 0.01)
    if dreamz:
        e.set_proposal_dreamz(64, capacity=64 + 3 * T + 128, **dreamz)
        e.set_archive(None)
    else:
        e.set_proposal(**prop)
    e.init(truth + 0.05 * rng.standard_normal((N, d)))
    P = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    S = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    Ac = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    e.run(min(T, 100), P, S, Ac)
    dt = bc._timed(lambda: e.run(T, P, S, Ac))
    e.set_profiling(True)
    bc._timed(lambda: e.run(T, P, S, Ac))
    p = e.profile()
    acc = float(Ac.float().mean())
    e.close()
    return dict(name=name, evals_per_s=N * T / dt, acceptance=acc, ms={k: round(v, 3) for k, v in p.items() if k.startswith("ms_")})


def hier(name, d, ms, sl, prop, n_fine, randomize=False, noise=0):
    torch = bc._torch()
    lv = bc.levels(ms, d=d)
    e = bc._engine()(N, d, seed=9, n_levels=len(ms))
    e.set_prior(np.zeros(d), np.eye(d))
    rng = np.random.default_rng(1)
    for k, (A, y) in enumerate(lv):
        if noise == 1:
            e.set_level(k, A, y, 1, 0.01 * (1 + rng.random(len(y))))
        else:
            e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(**prop)
    e.set_subchains(sl, randomize=randomize) if randomize else e.set_subchains(sl)
    e.init(None)
    rows = e.rows_per_level(n_fine)
    outs = bc._level_buffers(rows, N, d)
    e.run_levels(max(1, n_fine // 10), outs)
    dt = bc._timed(lambda: e.run_levels(n_fine, outs))
    e.set_profiling(True)
    bc._timed(lambda: e.run_levels(n_fine, outs))
    p = e.profile()
    acc = [round(float(o[2].float().mean().item()), 3) for o in outs]
    e.close()
    return dict(name=name, evals_per_s=N * rows[0] / dt, acceptance=acc, ms={k: round(v, 3) for k, v in p.items() if k.startswith("ms_")})


CASES = [
    ("grw d16 m256", lambda: single("grw d16 m256", 16, 256, dict(kind=0, C_=np.eye(16), scaling=0.05))),
    ("grw d32 m256", lambda: single("grw d32 m256", 32, 256, dict(kind=0, C_=np.eye(32), scaling=0.05))),
    ("grw d64 m256", lambda: single("grw d64 m256", 64, 256, dict(kind=0, C_=np.eye(64), scaling=0.03))),
    ("pcn d64 m1024 adaptive", lambda: single("pcn d64 m1024 adaptive", 64, 1024, dict(kind=1, scaling=0.03, adaptive=True))),
    # round 5: 65 .. 128 parameters (tda_kernels_wide.h) next to the 64-parameter headline shape; flops per evaluation 2 m d
    ("am d64 m1024", lambda: single("am d64 m1024", 64, 1024, dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100))),
    ("am d128 m1024", lambda: single("am d128 m1024", 128, 1024, dict(kind=2, C_=1e-4 * np.eye(128), t0=100, period=100))),
    ("am d96 m1024", lambda: single("am d96 m1024", 96, 1024, dict(kind=2, C_=1e-4 * np.eye(96), t0=100, period=100))),
    ("grw d128 m1024", lambda: single("grw d128 m1024", 128, 1024, dict(kind=0, C_=np.eye(128), scaling=0.02))),
    ("pcn d128 m1024 adaptive", lambda: single("pcn d128 m1024 adaptive", 128, 1024, dict(kind=1, scaling=0.02, adaptive=True))),
    ("am d32 m256", lambda: single("am d32 m256", 32, 256, dict(kind=2, C_=1e-4 * np.eye(32), t0=100, period=100))),
    ("am d64 m1024 diag noise", lambda: single("am d64 m1024 diag noise", 64, 1024, dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100), noise=1)),
    ("am d64 m1024 dense prior", lambda: single("am d64 m1024 dense prior", 64, 1024, dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100), prior="dense")),
    ("grw d64 m256 dense noise", lambda: single("grw d64 m256 dense noise", 64, 256, dict(kind=0, C_=np.eye(64), scaling=0.03), noise=2)),
    ("dreamz d64 m256", lambda: single("dreamz d64 m256", 64, 256, None, T=192, dreamz=dict(delta=2, nCR=3, adaptive=True, period=64))),
    ("dreamz d32 m256", lambda: single("dreamz d32 m256", 32, 256, None, T=192, dreamz=dict(delta=2, nCR=3, adaptive=True, period=64))),
    ("dreamz d16 m64", lambda: single("dreamz d16 m64", 16, 64, None, T=192, dreamz=dict(delta=1, nCR=3))),
    ("da d32 64/512 pcn", lambda: hier("da d32 64/512 pcn", 32, (64, 512), [10], dict(kind=1, scaling=0.03), 100)),
    ("da d64 256/2048 pcn diag noise", lambda: hier("da d64 256/2048 pcn diag noise", 64, (256, 2048), [10], dict(kind=1, scaling=0.02), 100, noise=1)),
    ("da d64 512/2048 pcn (generic kernel)", lambda: hier("da d64 512/2048 pcn (generic kernel)", 64, (512, 2048), [10], dict(kind=1, scaling=0.02), 60)),
    ("da d64 256/2048 pcn randomised", lambda: hier("da d64 256/2048 pcn randomised", 64, (256, 2048), [10], dict(kind=1, scaling=0.02), 100, randomize=True)),
    ("mlda3 d64 256/512/2048 am (generic kernel)", lambda: hier("mlda3 d64 256/512/2048 am (generic kernel)", 64, (256, 512, 2048), [5, 3], dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100), 40)),
    ("mlda4 d64 128/256/512/2048 grw", lambda: hier("mlda4 d64 128/256/512/2048 grw", 64, (128, 256, 512, 2048), [4, 3, 2], dict(kind=0, C_=np.eye(64), scaling=0.02), 30)),
    ("da d128 256/2048 pcn", lambda: hier("da d128 256/2048 pcn", 128, (256, 2048), [10], dict(kind=1, scaling=0.02), 60)),
    ("da d128 256/2048 am", lambda: hier("da d128 256/2048 am", 128, (256, 2048), [10], dict(kind=2, C_=1e-4 * np.eye(128), t0=100, period=100), 60)),
    ("da d64 256/2048 pcn (same harness)", lambda: hier("da d64 256/2048 pcn (same harness)", 64, (256, 2048), [10], dict(kind=1, scaling=0.02), 60)),
    ("mlda3 d128 256/512/2048 am", lambda: hier("mlda3 d128 256/512/2048 am", 128, (256, 512, 2048), [5, 3], dict(kind=2, C_=1e-4 * np.eye(128), t0=100, period=100), 30)),
    ("mlda4 d128 128/256/512/2048 grw", lambda: hier("mlda4 d128 128/256/512/2048 grw", 128, (128, 256, 512, 2048), [4, 3, 2], dict(kind=0, C_=np.eye(128), scaling=0.02), 20)),
    ("mlda5 d64 64/128/256/512/2048 pcn", lambda: hier("mlda5 d64 64/128/256/512/2048 pcn", 64, (64, 128, 256, 512, 2048), [3, 2, 2, 2], dict(kind=1, scaling=0.02), 16)),
    ("mlda6 d32 32/64/128/256/512/1024 am", lambda: hier("mlda6 d32 32/64/128/256/512/1024 am", 32, (32, 64, 128, 256, 512, 1024), [2, 2, 2, 2, 2], dict(kind=2, C_=1e-4 * np.eye(32), t0=100, period=100), 10)),
    ("mlda3 d32 64/256/1024 am", lambda: hier("mlda3 d32 64/256/1024 am", 32, (64, 256, 1024), [5, 3], dict(kind=2, C_=1e-4 * np.eye(32), t0=100, period=100), 40)),
]

if __name__ == "__main__":
    if "--lib" in sys.argv:  # A/B against another build of the library:  --lib path/to/libtinyda_hip.so
        from tinyda_amd import _lib

        i = sys.argv.index("--lib")
        _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, fn in CASES:
        if flt not in name:
            continue
        try:
            r = fn()
        except Exception as exc:  # noqa: BLE001
            r = dict(name=name, error=repr(exc))
        print(json.dumps(r), flush=True)
