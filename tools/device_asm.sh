#!/bin/bash
# device assembly of the engine (gfx950) -> $1 (default /tmp/tda_dev.s); same flags as __graft_entry__.build()
OUT=${1:-/tmp/tda_dev.s}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -S -I$ROOT/include -I$ROOT/tinyda_amd/csrc \
  -o $OUT $ROOT/tinyda_amd/csrc/tda_engine.hip
