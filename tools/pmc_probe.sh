# counters of a stand-alone probe binary: bash tools/pmc_probe.sh OUT.json "COUNTER ..." binary [args]   (one --pmc pass)
OUT=$1; C=$2; shift 2
cd /tmp && export TMPDIR=/tmp
P=/tmp/pmcprobe_$$
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P -- "$@" > $P.log 2>&1)
python3 - $P $GRAFT_REPO_ROOT/gpurun_out/$OUT <<'PY'
import csv, glob, json, statistics as st, collections, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: dict({c: st.median(v) for c, v in cs.items()}, launches=len(next(iter(cs.values())))) for k, cs in per.items()}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
