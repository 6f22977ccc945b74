# round-4 profile pass on the GPU box (after tools/r04_measure.sh): kernel stats of the driver's bench command with the configs block,
# HBM traffic of the headline kernels, SQ + traffic counters of every other configuration's kernels.  Files land in gpurun_out/.
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04}
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/prof_${TAG}.log 2>&1; find /tmp/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv \; ; grep '^{"metric"' /tmp/prof_${TAG}.log > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json)
echo "kernel stats done" | tee -a gpurun_out/${TAG}_progress.log
PMC_OUT=${TAG}_pmc_traffic.json bash tools/pmc_traffic.sh > gpurun_out/${TAG}_pmc.log 2>&1; echo "pmc traffic rc=$?" | tee -a gpurun_out/${TAG}_progress.log
bash tools/pmc_any.sh ${TAG}_pmc_c3.json python3 tools/bench_configs.py c3 > gpurun_out/${TAG}_pmc_c3.log 2>&1; echo "c3 rc=$?" | tee -a gpurun_out/${TAG}_progress.log
bash tools/pmc_any.sh ${TAG}_pmc_c5.json python3 tools/bench_configs.py c5 > gpurun_out/${TAG}_pmc_c5.log 2>&1; echo "c5 rc=$?" | tee -a gpurun_out/${TAG}_progress.log
bash tools/pmc_any.sh ${TAG}_pmc_c4.json python3 tools/bench_configs.py c4 16 > gpurun_out/${TAG}_pmc_c4.log 2>&1; echo "c4 rc=$?" | tee -a gpurun_out/${TAG}_progress.log
bash tools/pmc_any.sh ${TAG}_pmc_c5aem.json python3 tools/bench_configs.py c5aem 128 > gpurun_out/${TAG}_pmc_c5aem.log 2>&1; echo "c5aem rc=$?" | tee -a gpurun_out/${TAG}_progress.log
bash tools/pmc_any.sh ${TAG}_pmc_c2b.json python3 tools/bench_configs.py c2b > gpurun_out/${TAG}_pmc_c2b.log 2>&1; echo "c2b rc=$?" | tee -a gpurun_out/${TAG}_progress.log
