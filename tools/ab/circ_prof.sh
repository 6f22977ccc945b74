#!/bin/bash
# per-kernel times of the two variants of the moment recursion (rocprofv3 --kernel-trace --stats), same box
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/circ
for v in 0 1; do
  export TINYDA_ADAPT_CIRC=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/circ/prof_$v -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ess --no-configs > gpurun_out/circ/prof_$v.log 2>&1
  f=$(find gpurun_out/circ/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== circ=$v $f"
  if [ -n "$f" ]; then head -8 "$f" | cut -c1-200; cp "$f" gpurun_out/circ/kernel_stats_$v.csv; fi
  rm -rf gpurun_out/circ/prof_$v
done
