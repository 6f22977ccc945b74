# Matrix-core counters behind every `frac`: bash tools/pmc_mfma.sh [OUT.json]
#   one --pmc pass (kernel-trace only, as gpurun requires) per configuration:
#     SQ_VALU_MFMA_BUSY_CYCLES  SQ_INSTS_VALU_MFMA_MOPS_F64  SQ_BUSY_CU_CYCLES  GRBM_GUI_ACTIVE
#   over  C2a (bench.py, k_mh_steps<64,8>), C2b (k_mh_steps<64,4>), C3 and C5-literal (k_da_steps_r224), C5 + dense error model
#   (k_aem_refresh, k_aem_action, k_aem_base_steps).
# Per kernel (median over the upper half of its launches, like tools/pmc_any.sh):
#   mfma_flops_counted = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512   (counter_defs.yaml: one MOP = 512 flop; a v_mfma_f64_16x16x4 is 4)
#   mfma_util_counter  = mfma_flops_counted / launch duration / 78.6 TFLOP/s      (what the matrix cores DID, padding and all)
#   mfma_busy_frac     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
# and, where the run's JSON line gives the evaluations per launch, counted_over_algorithmic = counted / (flops per eval x evals):
# above 1 = padded or redundant matrix work the algorithmic count hides, below 1 = work done on the vector unit.
# The output goes to gpurun_out/; review, then copy to profiles/rNN_pmc_mfma.json (bench.py reads the newest one).
OUT=${1:-r05_pmc_mfma.json}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export TINYDA_CONFIGS_REPS=1 TINYDA_CONFIGS_WINDOW_S=0
P=/tmp/pmcmfma_$$
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
i=0
for cfg in "bench" "c2b" "c3" "c5" "c5aem 128"; do
  i=$((i+1))
  if [ "$cfg" = "bench" ]; then
    (cd $ROOT && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P$i -- python3 bench.py --steps 10 --warmup 2 --pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess --no-configs > $P$i.log 2>&1)
  else
    (cd $ROOT && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P$i -- python3 tools/bench_configs.py $cfg > $P$i.log 2>&1)
  fi
  echo "[pmc_mfma] pass $i ($cfg) done: $(tail -c 300 $P$i.log | tr '\n' ' ' | cut -c1-200)"
done
python3 - $P $ROOT $ROOT/gpurun_out/$OUT <<'PY'
import collections, csv, glob, hashlib, json, os, statistics as st, sys
P, ROOT, OUT = sys.argv[1:4]
PEAK = 78.6e12
cfgs = ["C2a", "C2b", "C3", "C5-literal", "C5+AEM128"]
kernels = {}
lines = {}
for i, tag in enumerate(cfgs, 1):
    # the run's own JSON line: flops per evaluation and evaluations per launch of the step kernel
    try:
        for ln in open("%s%d.log" % (P, i)):
            if ln.startswith("{"):
                lines[tag] = json.loads(ln)
    except Exception:
        pass
    fs = glob.glob("%s%d/**/*counter_collection.csv" % (P, i), recursive=True)
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        if "tda::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0]
        did = r.get("Dispatch_Id") or r.get("Correlation_Id")
        per[k][did][r["Counter_Name"]] = per[k][did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k][did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if not any(dur.values()):  # older column set: durations from the kernel trace of the same pass
        for f in glob.glob("%s%d/**/*kernel_trace.csv" % (P, i), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if k in per:
                    dur[k][r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    for k, disp in per.items():
        # the block launches are the big ones: the upper half by matrix-core work (by duration when a kernel has none)
        ids = sorted(disp, key=lambda d: (disp[d].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0), dur[k].get(d, 0.0)))
        ids = ids[len(ids) // 2:]
        med = lambda name: st.median([disp[d].get(name, 0.0) for d in ids])
        e = {"config": tag, "launches": len(disp)}
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"):
            e[c] = med(c)
        ds = [dur[k][d] for d in ids if d in dur[k]]
        e["duration_ms"] = st.median(ds) * 1e3 if ds else None
        e["mfma_flops_counted"] = e["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
        e["mfma_util_counter"] = e["mfma_flops_counted"] / (e["duration_ms"] * 1e-3) / PEAK if e["duration_ms"] else None
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if e["GRBM_GUI_ACTIVE"] else None
        kernels["%s | %s" % (tag, k)] = e

def blob(rel):
    data = open(os.path.join(ROOT, rel), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()

out = {"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE, one pass per "
               "configuration (tools/pmc_mfma.sh); per kernel the median over the upper half of its launches; mfma_flops_counted = MOPS_F64 x 512; "
               "mfma_util_counter = counted flops / launch duration (same pass) / 78.6 TFLOP/s; mfma_busy_frac = MFMA busy cycles / "
               "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).",
       "source_blobs": {f: blob(os.path.join("tinyda_amd", "csrc", f)) for f in
                        ("tda_kernels_mh.h", "tda_kernels_ml.h", "tda_kernels_da_body.inc", "tda_kernels_aemr.h")},
       "kernels": kernels}
# the headline kernel: counted against algorithmic flops per 100-step launch of 4096 chains
FLOPS_PER_EVAL_C2A = 2 * 1024 * 64 + 3 * 1024 + 2 * 64
for name, e in kernels.items():
    if name.startswith("C2a") and "k_mh_steps<64, 8" in name:
        alg = FLOPS_PER_EVAL_C2A * 409600.0
        out["k_mh_steps"] = {"kernel": name, "evals_per_launch": 409600, "algorithmic_flops_per_launch": alg,
                             "mfma_flops_counted_per_launch": e["mfma_flops_counted"], "counted_over_algorithmic": e["mfma_flops_counted"] / alg,
                             "mfma_util_counter": e["mfma_util_counter"], "mfma_busy_frac": e["mfma_busy_frac"], "duration_ms": e["duration_ms"]}
# the other configurations: algorithmic flops per step-kernel launch from the run's own line
for tag, ln in lines.items():
    if tag == "C2a" or "flops_or_bytes_per_eval" not in ln:
        continue
    per_launch = ln.get("evals_per_steps_launch")
    for name, e in kernels.items():
        if name.startswith(tag + " |") and per_launch and ("k_mh_steps" in name or "k_da_steps" in name):
            e["algorithmic_flops_per_launch"] = ln["flops_or_bytes_per_eval"] * per_launch
            e["counted_over_algorithmic"] = e["mfma_flops_counted"] / e["algorithmic_flops_per_launch"]
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps({k: {x: v.get(x) for x in ("launches", "duration_ms", "mfma_util_counter", "mfma_busy_frac", "counted_over_algorithmic")}
                  for k, v in kernels.items() if v["SQ_INSTS_VALU_MFMA_MOPS_F64"] > 0}, indent=1))
PY
