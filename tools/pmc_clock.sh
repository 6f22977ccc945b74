# effective clock per kernel of the headline run (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE / 8 / wall time of the dispatch
# (rocprofv3 reports the sum over the 8 XCDs; reads high on dispatches much shorter than 0.3 ms).  Counter collection serialises
# the two streams, so k_rng does not run next to k_mh_steps here.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcclk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess --no-configs > /tmp/pmcclk.log 2>&1
python3 - <<'PY'
import csv, glob, json, statistics as st, collections
f = glob.glob("/tmp/pmcclk/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
dur = {}
if rows and "Start_Timestamp" not in rows[0]:
    t = glob.glob("/tmp/pmcclk/**/*kernel_trace.csv", recursive=True)[0]
    for r in csv.DictReader(open(t)):
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
per = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "tda::" not in r["Kernel_Name"]:
        continue
    ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) if "Start_Timestamp" in r else dur.get(r["Dispatch_Id"], 0.0)
    if ns > 0:
        per[r["Kernel_Name"].split("(")[0]].append((ns, float(r["Counter_Value"]) / 8.0 / ns))
out = {}
for k, v in per.items():
    v.sort()
    big = v[len(v) // 2:]  # the block launches
    out[k] = {"dispatches": len(v), "median_us": round(st.median(x[0] for x in big) / 1e3, 1), "effective_clock_GHz": round(st.median(x[1] for x in big), 3)}
json.dump(out, open("/tmp/pmc_clock.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp /tmp/pmc_clock.json $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_clock.json
