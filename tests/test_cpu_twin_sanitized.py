"""oracle/tda_cpu_abi.cpp -- the CPU twin of the C-ABI, 500 lines of index arithmetic shared in spirit with the HIP engine's host
side -- built with AddressSanitizer + UndefinedBehaviourSanitizer (CPU only: GPU sanitizers are not available on this pool).
tests/test_cpu_abi.py (five reference traces replayed through the Engine wrapper, split runs, the Philox forward mode, the
capacity contract, every exported symbol) is re-run in a child process against that build: any out-of-bounds access,
use-after-free, signed overflow or misaligned access aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_twin_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    so = str(tmp_path / "libtda_cpu_san.so")
    cc = subprocess.run(["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                         "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas",
                         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tinyda_amd", "csrc"), "-o", so,
                         os.path.join(ROOT, "oracle", "tda_cpu_abi.cpp"), "-lm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert cc.returncode == 0, cc.stdout
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(libasan), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="3", TINYDA_CPU_ABI_SO=so)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_cpu_abi.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    import re

    n = re.search(r"(\d+) passed", r.stdout)
    assert n and int(n.group(1)) >= 7 and "skipped" not in r.stdout, r.stdout[-2000:]
