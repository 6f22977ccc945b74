# SQ counters of any command's tda kernels: bash tools/pmc_any.sh OUT.json python3 tools/bench_configs.py c4 16
# (two separate --pmc passes, kernel-trace only, as gpurun requires; median over the upper half of each kernel's launches)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
P=/tmp/pmcany_$$
for pass in 1 2 3; do
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; elif [ $pass = 2 ]; then C="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; else C="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; fi
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P$pass -- "$@" > $P$pass.log 2>&1)
done
python3 - $P $GRAFT_REPO_ROOT/gpurun_out/$OUT <<'PY'
import csv, glob, json, statistics as st, collections, sys
res = collections.defaultdict(dict)
for p in (1, 2, 3):
    fs = glob.glob("%s%d/**/*counter_collection.csv" % (sys.argv[1], p), recursive=True)
    if not fs: continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "tda::" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            v = sorted(v)
            res[k][c] = st.median(v[len(v) // 2:])
            res[k]["launches"] = len(v)
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
