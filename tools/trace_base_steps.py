#!/usr/bin/env python3
"""Per-wave cycle stamps of k_aem_base_steps on the C5 + dense error model configuration (4096 chains, 128 outputs).

Build a library with the stamps compiled in and run this on the GPU box:

    hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -Iinclude -Itinyda_amd/csrc \\
          -DAEMB_TRACE tinyda_amd/csrc/tda_engine.hip -L/opt/rocm/lib -lhipfft -lhiprtc -o tools/bin/libs/libtda_trace.so
    python tools/trace_base_steps.py tools/bin/libs/libtda_trace.so

The 40th launch of the kernel writes [chain][16] s_memtime stamps (wave start, staging done, block rows 0..7 done, steps start,
steps done) to AEMB_TRACE_FILE; the counters of different XCDs are not aligned, so only differences inside a wave are reported.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    lib = os.path.abspath(sys.argv[1])
    out = os.environ.setdefault("AEMB_TRACE_FILE", "/tmp/aemb_trace.bin")
    from tinyda_amd import _lib

    _lib.LIB_PATH = lib
    import bench_configs as bc

    bc.REPS = 1
    r = bc.run_c5_aem(m=128, n_fine=20)
    print("coarse evals/s (stamps compiled in):", r["evals_per_s"])
    t = np.fromfile(out, dtype=np.uint64).reshape(8192, 16)[:4096, :12].astype(np.int64)
    dur = np.diff(t, axis=1)
    names = ["staging"] + [f"row {q}" for q in range(8)] + ["to steps", "steps"]
    print("cycles per wave, start to end: median", int(np.median(t[:, 11] - t[:, 0])))
    for j, n in enumerate(names):
        print(f"  {n:9s} median {int(np.median(dur[:, j])):7d}  p10 {int(np.percentile(dur[:, j], 10)):7d}  p90 {int(np.percentile(dur[:, j], 90)):7d}")


if __name__ == "__main__":
    main()
