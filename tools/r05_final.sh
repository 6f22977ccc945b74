# round-5 final evidence pass, in parts that each fit one gpurun call (PART=A|B|C|D); outputs under gpurun_out/, copied into profiles/ by hand
cd $GRAFT_REPO_ROOT
case "${PART:-A}" in
A)  # GPU tests, the driver's bench command, its rocprofv3 kernel stats, the matrix-core counters
  TAG=r05 PMC=1 bash tools/r05_measure.sh ;;
B)  # HBM traffic of the headline kernels; SQ / traffic counters of C4 and of C5 + dense error model
  PMC_OUT=r05_pmc_traffic.json bash tools/pmc_traffic.sh > gpurun_out/r05_pmc_traffic.log 2>&1; echo "traffic rc=$?"
  bash tools/pmc_any.sh r05_pmc_c4.json python3 tools/bench_configs.py c4 16 > gpurun_out/r05_pmc_c4.log 2>&1; echo "c4 rc=$?"
  bash tools/pmc_any.sh r05_pmc_c5aem.json python3 tools/bench_configs.py c5aem 128 > gpurun_out/r05_pmc_c5aem.log 2>&1; echo "c5aem rc=$?" ;;
C)  # rates: 65 .. 128 parameters (single level and hierarchies), C5 + dense error model at 128 / 256 outputs with kernel stats, the refresh probe
  for f in "am d64 m1024" "am d128" "am d96" "grw d128" "pcn d128" "da d128" "da d64 256/2048 pcn (same" "mlda3 d128" "mlda4 d128" "mlda3 d64" "mlda4 d64" "mlda5 d64" "mlda6 d32"; do python tools/rate_sweep.py "$f"; done > gpurun_out/r05_rate_wide.jsonl 2> gpurun_out/r05_rate_wide.err; echo "rates rc=$?"
  python tools/bench_configs.py c5aem 128 > gpurun_out/r05_c5aem_m128.json 2> gpurun_out/r05_c5aem.err; echo "c5aem128 rc=$?"
  python tools/bench_configs.py c5aem 256 10 > gpurun_out/r05_c5aem_m256.json 2>> gpurun_out/r05_c5aem.err; echo "c5aem256 rc=$?"
  (cd /tmp && export TMPDIR=/tmp && TINYDA_CONFIGS_REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5aem -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py c5aem 128 > /tmp/prof_c5aem.log 2>&1; find /tmp/prof_c5aem -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05_c5aem_kernel_stats.csv \; )
  (cd /tmp && export TMPDIR=/tmp && TINYDA_CONFIGS_REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5aem256 -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py c5aem 256 10 > /tmp/prof_c5aem256.log 2>&1; find /tmp/prof_c5aem256 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05_c5aem_m256_kernel_stats.csv \; )
  echo "stats done"
  [ -x tools/bin/arp ] && { timeout -k 10 200 tools/bin/arp 4096 > gpurun_out/r05_aem_refresh_probe.txt 2>&1; timeout -k 10 200 tools/bin/arp 4096 big >> gpurun_out/r05_aem_refresh_probe.txt 2>&1; echo "probe rc=$?"; }
  [ -f tools/bin/libs/libtda_trace.so ] && { python tools/trace_base_steps.py tools/bin/libs/libtda_trace.so > gpurun_out/r05_base_steps_trace.txt 2>&1; echo "trace rc=$?"; } ;;
D)  # the extended random sweep as a test
  TINYDA_SWEEP=${SWEEP:-740} python -m pytest tests/test_gpu_sweep.py -q -m gpu -k extended > gpurun_out/r05_sweep.log 2>&1; echo "sweep rc=$? $(tail -1 gpurun_out/r05_sweep.log)" ;;
esac
echo finished
