#!/bin/bash
# same-box A/B of the circulant diagonal blocks in the moment recursion (TINYDA_ADAPT_CIRC=0 / 1): switch test, then alternating bench runs
set -e
mkdir -p gpurun_out/circ
rm -f gpurun_out/circ/ab.txt
timeout -k 10 500 python -m pytest tests/test_gpu_switches.py -x -q -k "fused_swap" > gpurun_out/circ/switch.log 2>&1 || { tail -30 gpurun_out/circ/switch.log; exit 1; }
tail -2 gpurun_out/circ/switch.log
for i in 1 2 3; do
  for v in 0 1; do
    TINYDA_ADAPT_CIRC=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ess --no-configs 2>/dev/null | tail -1 > gpurun_out/circ/line_$v.json
    python -c "import sys,json; d=json.load(open('gpurun_out/circ/line_$v.json')); print('circ=$v', d['value'], d['ms_per_step'])" >> gpurun_out/circ/ab.txt
  done
done
cat gpurun_out/circ/ab.txt
