# SQ counters of any command's tda kernels: bash tools/pmc_any.sh OUT.json python3 tools/bench_configs.py c4 16
# (separate --pmc passes, kernel-trace only, as gpurun requires: three of SQ counters, two of HBM traffic -- FETCH_SIZE | WRITE_SIZE +
# TCC hits / misses; median over the upper half of each kernel's launches; hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE
# doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
P=/tmp/pmcany_$$
for pass in 1 2 3 4 5; do
  if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; elif [ $pass = 2 ]; then C="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; elif [ $pass = 3 ]; then C="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; elif [ $pass = 4 ]; then C="FETCH_SIZE"; else C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; fi
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P$pass -- "$@" > $P$pass.log 2>&1)
done
python3 - $P $GRAFT_REPO_ROOT/gpurun_out/$OUT <<'PY'
import csv, glob, json, statistics as st, collections, sys
res = collections.defaultdict(dict)
for p in (1, 2, 3, 4, 5):
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
for k, c in res.items():
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        c["hbm_bytes_corrected"] = int((2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024)
        c["L2_hit_rate"] = c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
