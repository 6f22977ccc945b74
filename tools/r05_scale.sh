#!/bin/bash
# One command for the 8-GPU node:   bash tools/r05_scale.sh [OUT_DIR]
#   1. the headline bench at --gpus 1, 2, 4, 8 (weak scaling, no data-path collective) with efficiency = value_N / (N * value_1)
#   2. BASELINE config 4 (DREAM shared archive) at 1 and NG ranks in the four archive modes
#   3. tools/peer_archive_check.py --nproc-per-node NG (distributed archive against the replicated one, across devices)
#   4. the other BASELINE configurations of bench.py's `configs` block (C2b, C3, C5-literal, C5 + dense error model) per rank count,
#      weak scaling, no collective (tools/configs_scale.py)
# Every output line carries rccl_ranks; per-rank rates are in the C4 lines.  REHEARSE=1: two ranks on ONE GPU over gloo with small
# sizes (the code path, not a measurement) -- tests/test_gpu_bench.py runs it that way.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/r05_scale}
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
if [ "${REHEARSE:-0}" = "1" ]; then
  export TINYDA_BENCH_ONE_GPU=1
  GPUS="1 2"; NG=2
  BENCH_ARGS="--steps 3 --warmup 1 --chains 1024 --pilot 200 --burnin 500 --ess-iterations 0 --no-cpu-baseline"
  C4_ARGS="--chains 512 --steps 64 --sync 16"
  CFG_ARGS="--small"
else
  GPUS=""; for n in 1 2 4 8; do [ "$n" -le "$NDEV" ] && GPUS="$GPUS $n"; done
  NG=$(echo $GPUS | awk '{print $NF}')
  BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline"
  C4_ARGS="--chains 8192 --steps 400 --sync 16"
  CFG_ARGS=""
fi
echo "[r05_scale] devices visible: $NDEV; bench at --gpus:$GPUS; C4 / peer check with $NG ranks; output in $OUT"
port() { python3 -c "import socket; s = socket.socket(); s.bind(('127.0.0.1', 0)); print(s.getsockname()[1])"; }
rc=0
: > "$OUT/bench_scale.jsonl"
for n in $GPUS; do
  python3 bench.py --gpus $n $BENCH_ARGS >> "$OUT/bench_scale.jsonl" 2> "$OUT/bench_gpus$n.err" || { echo "[r05_scale] bench --gpus $n FAILED (see $OUT/bench_gpus$n.err)"; rc=1; }
done
: > "$OUT/c4_scale.jsonl"
for n in 1 $NG; do
  for mode in replicated-blocking replicated-overlapped distributed-sync distributed-lagged; do
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $(port) tools/c4_scale.py --mode $mode $C4_ARGS \
      >> "$OUT/c4_scale.jsonl" 2> "$OUT/c4_${mode}_$n.err" || { echo "[r05_scale] c4 $mode with $n ranks FAILED (see $OUT/c4_${mode}_$n.err)"; rc=1; }
  done
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $NG --master-addr 127.0.0.1 --master-port $(port) tools/peer_archive_check.py \
  > "$OUT/peer_archive_check.jsonl" 2> "$OUT/peer_archive_check.err" || { echo "[r05_scale] peer_archive_check FAILED (see $OUT/peer_archive_check.err)"; rc=1; }
: > "$OUT/configs_scale.jsonl"
for n in $GPUS; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $(port) tools/configs_scale.py $CFG_ARGS \
    >> "$OUT/configs_scale.jsonl" 2> "$OUT/configs_$n.err" || { echo "[r05_scale] configs with $n ranks FAILED (see $OUT/configs_$n.err)"; rc=1; }
done
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
def lines(p):
    try:
        return [json.loads(l) for l in open(p) if l.startswith("{")]
    except OSError:
        return []
rep = {"bench": [], "c4": [], "peer_archive_check": lines(out + "/peer_archive_check.jsonl")}
b = lines(out + "/bench_scale.jsonl")
one = next((x["value"] for x in b if x["n_gpus"] == 1), None)
for x in b:
    rep["bench"].append({"n_gpus": x["n_gpus"], "value": x["value"], "unit": x["unit"], "ms_per_step": x["ms_per_step"], "rccl_ranks": x["config"]["rccl_ranks"],
                         "per_rank_evals_per_s": x["value"] / x["n_gpus"], "efficiency": (x["value"] / (x["n_gpus"] * one)) if one else None})
c = lines(out + "/c4_scale.jsonl")
base = {x["mode"]: x["evals_per_s"] for x in c if x["n_gpus"] == 1}
for x in c:
    x["efficiency"] = x["evals_per_s"] / (x["n_gpus"] * base[x["mode"]]) if x["mode"] in base else None
    rep["c4"].append(x)
cf = lines(out + "/configs_scale.jsonl")
base_cf = {x["tag"]: x["evals_per_s"] for x in cf if x["n_gpus"] == 1}
for x in cf:
    x["efficiency"] = x["evals_per_s"] / (x["n_gpus"] * base_cf[x["tag"]]) if x["tag"] in base_cf else None
rep["configs"] = cf
json.dump(rep, open(out + "/summary.json", "w"), indent=1)
for x in rep["bench"]:
    print("[r05_scale] bench --gpus %d: %.4g %s, efficiency %s, rccl_ranks %s" % (x["n_gpus"], x["value"], x["unit"], "%.3f" % x["efficiency"] if x["efficiency"] else "n/a", x["rccl_ranks"]))
for x in rep["c4"]:
    print("[r05_scale] C4 %-22s %d rank(s): %.4g evals/s, efficiency %s, rccl_ranks %s%s" % (x["mode"], x["n_gpus"], x["evals_per_s"], "%.3f" % x["efficiency"] if x["efficiency"] else "n/a", x["rccl_ranks"], (" -- " + x["note"]) if x.get("note") else ""))
for x in rep["configs"]:
    print("[r05_scale] %-12s %d rank(s): %.4g evals/s, efficiency %s, rccl_ranks %s" % (x["tag"], x["n_gpus"], x["evals_per_s"], "%.3f" % x["efficiency"] if x["efficiency"] else "n/a", x["rccl_ranks"]))
PY
[ -s "$OUT/summary.json" ] || rc=1
exit $rc
