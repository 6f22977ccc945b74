# round-5 measurement pass on the GPU box: tests, the driver's bench command (with the `configs` block), rocprofv3 kernel stats of it,
# and (PMC=1) the matrix-core counter passes.  TAG names the outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05}
if [ "${TESTS:-1}" = "1" ]; then
  python -u -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/${TAG}_gputests.log)" | tee -a gpurun_out/${TAG}_progress.log
fi
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" | tee -a gpurun_out/${TAG}_progress.log
if [ "${PROF:-1}" = "1" ]; then
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_${TAG}.log 2>&1; find /tmp/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv \; ; grep '^{"metric"' /tmp/prof_${TAG}.log | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json)
echo "rocprof done" | tee -a gpurun_out/${TAG}_progress.log
fi
if [ "${PMC:-0}" = "1" ]; then
  bash tools/pmc_mfma.sh ${TAG}_pmc_mfma.json > gpurun_out/${TAG}_pmc_mfma.log 2>&1; echo "pmc_mfma rc=$?" | tee -a gpurun_out/${TAG}_progress.log
fi
