# round-3 measurement pass on the GPU box: tests, the driver's bench command, rocprofv3 kernel stats of it, PMC traffic, configs, API
cd $GRAFT_REPO_ROOT
python -u -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r03_progress.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r03_progress.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_r03.log 2>&1; find /tmp/prof_r03 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r03_kernel_stats.csv \; ; grep '^{"metric"' /tmp/prof_r03.log | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_under_rocprof.json)
echo "rocprof done" | tee -a gpurun_out/r03_progress.log
PMC_OUT=r03_pmc_traffic.json bash tools/pmc_traffic.sh > gpurun_out/r03_pmc.log 2>&1; echo "pmc rc=$?" | tee -a gpurun_out/r03_progress.log
python -u tools/api_rate.py 2000 6 > gpurun_out/r03_api.json 2> gpurun_out/r03_api.err; echo "api rc=$?" | tee -a gpurun_out/r03_progress.log
python -u tools/bench_configs.py > gpurun_out/r03_configs.jsonl 2>&1; echo "configs rc=$?" | tee -a gpurun_out/r03_progress.log
python -u tools/bench_configs.py c4 128 >> gpurun_out/r03_configs.jsonl 2>&1
python -u tools/bench_configs.py c5aem 128 >> gpurun_out/r03_configs.jsonl 2>&1
python -u tools/bench_configs.py c5aemd 128 >> gpurun_out/r03_configs.jsonl 2>&1
python -u tools/bench_configs.py c4peer 16 >> gpurun_out/r03_configs.jsonl 2>&1
python -u tools/bench_configs.py c4peerlag 16 >> gpurun_out/r03_configs.jsonl 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --extras > gpurun_out/r03_bench_extras.json 2> gpurun_out/r03_bench_extras.err
echo "all done" | tee -a gpurun_out/r03_progress.log
