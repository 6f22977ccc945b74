cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  if [ $pass = 1 ]; then C="FETCH_SIZE"; else C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc$pass -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess --no-configs > /tmp/pmc$pass.log 2>&1
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
python3 - <<'PY'
# the file bench.py reads (profiles/rNN_pmc_traffic.json after review): per-launch figure of the dominant kernel + the git blob of
# the kernel source it was measured on (bench.py prints traffic_stale when the source has changed since)
import hashlib, json, os
root = os.environ["GRAFT_REPO_ROOT"]
raw = json.load(open("/tmp/pmc_out.json"))
data = open(os.path.join(root, "tinyda_amd", "csrc", "tda_kernels_mh.h"), "rb").read()
key = [k for k in raw if "k_mh_steps<64, 8" in k][0]
out = {"note": "rocprofv3 --pmc, two separate passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum) of `bench.py --steps 10 --warmup 2 "
               "--pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess --no-configs` (tools/pmc_traffic.sh); median over the 100-iteration launches "
               "(409600 evals each). FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced streaming reads on gfx950; "
               "units KiB -> bytes.",
       "evals_per_launch": 409600, "k_mh_steps_hbm_bytes_per_launch": raw[key]["hbm_bytes_corrected"],
       "k_mh_steps_algorithmic_bytes_per_launch": 1057 * 409600,
       "kernel_source_blob": hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest(), "kernels": raw}
json.dump(out, open(os.path.join(root, "gpurun_out", os.environ.get("PMC_OUT", "r03_pmc_traffic.json")), "w"), indent=1)
PY
