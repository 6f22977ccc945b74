cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  if [ $pass = 1 ]; then C="FETCH_SIZE"; else C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc$pass -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess > /tmp/pmc$pass.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, statistics as st, collections
res = collections.defaultdict(dict)
for p in (1, 2):
    f = glob.glob("/tmp/pmc%d/**/*counter_collection.csv" % p, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        if "tda::" not in k: continue
        for c, v in cs.items():
            v = sorted(v)
            # the block launches are the big ones: median of the upper half
            res[k][c] = st.median(v[len(v) // 2:])
            res[k]["n_" + c] = len(v)
out = {}
for k, c in res.items():
    f, w = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    out[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_corrected": int((2 * f + w) * 1024),
              "L2_hit_rate": hit / max(hit + miss, 1.0), "dispatches_seen": c.get("n_FETCH_SIZE", 0)}
json.dump(out, open("/tmp/pmc_out.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp /tmp/pmc_out.json $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_raw.json
