"""The C oracle built with AddressSanitizer + UndefinedBehaviourSanitizer (CPU only: GPU sanitizers are not available on
this pool) replays reference traces in a child process: any out-of-bounds access, use-after-free, signed overflow or
misaligned access in oracle/oracle_mh.c aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle_c
oracle_c._SO = %(so)r
oracle_c.load()
for name, nthreads in (("g2_am_small", 1), ("g2_am_small_adaptive", 3), ("g2_am_c2", 2)):
    g = np.load(os.path.join(%(root)r, "tests", "golden", name + ".npz"))
    res = oracle_c.run_mh(g["A"], g["data"], float(g["noise_cov"][0]), g["prior_mean"], np.diag(g["prior_cov"]), 2, g["C0"],
                          g["theta0"], np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1), adaptive=bool(g["adaptive"]),
                          gamma=float(g["gamma"]), period=int(g["period"]), sd=float(g["sd"]), eps=float(g["epsilon"]),
                          t0=int(g["t0"]), n_threads=nthreads)
    assert np.array_equal(res["accepted"], np.swapaxes(g["accepted"][:, 1:], 0, 1)), name
# ragged sizes, records off
rng = np.random.default_rng(0)
for (N, d, m, T) in ((1, 1, 1, 3), (5, 3, 7, 33), (2, 17, 5, 21)):
    A = rng.standard_normal((m, d))
    oracle_c.run_mh(A, rng.standard_normal(m), 0.3, np.zeros(d), np.ones(d), 2, 0.1 * np.eye(d), rng.standard_normal((N, d)),
                    rng.standard_normal((T, N, d)), rng.random((T, N)), period=10, t0=5, n_threads=2, want_records=False)
print("SANITIZED_OK")
"""


def test_c_oracle_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    so = str(tmp_path / "liboracle_mh_san.so")
    cc = subprocess.run(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                         "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", so,
                         os.path.join(ROOT, "oracle", "oracle_mh.c"), "-lm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert cc.returncode == 0, cc.stdout
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(libasan), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="3")
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, so=so)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, r.stdout[-4000:]
