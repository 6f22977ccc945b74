# round-4 measurement pass on the GPU box: tests, the driver's bench command (with the `configs` block), rocprofv3 kernel stats of it
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04}
python -u -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${TAG}_progress.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" | tee -a gpurun_out/${TAG}_progress.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_${TAG}.log 2>&1; find /tmp/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv \; ; grep '^{"metric"' /tmp/prof_${TAG}.log | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json)
echo "rocprof done" | tee -a gpurun_out/${TAG}_progress.log
