# SQ counters of the headline kernels (issue and wait cycles per wave): two separate --pmc passes over a short bench run
cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; else C="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcsq$pass -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess > /tmp/pmcsq$pass.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, statistics as st, collections
res = collections.defaultdict(dict)
for p in (1, 2):
    f = glob.glob("/tmp/pmcsq%d/**/*counter_collection.csv" % p, recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "tda::" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            v = sorted(v)
            res[k][c] = st.median(v[len(v) // 2:])  # the block launches are the big ones
json.dump(res, open("/tmp/pmc_sq.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cp /tmp/pmc_sq.json $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_sq.json
