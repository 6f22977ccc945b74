#!/bin/bash
# SQ counters of the period-boundary kernel (three --pmc passes of a short headline run)
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 10 --warmup 2 --pilot 1000 --burnin 2000 --no-cpu-baseline --no-ess --no-configs"
timeout -k 10 200 bash tools/pmc_probe.sh adapt_pmc1.json "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $B > /dev/null 2>&1; echo p1 $?
timeout -k 10 200 bash tools/pmc_probe.sh adapt_pmc2.json "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" $B > /dev/null 2>&1; echo p2 $?
timeout -k 10 200 bash tools/pmc_probe.sh adapt_pmc3.json "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" $B > /dev/null 2>&1; echo p3 $?
python3 - <<'PY'
import json
for i in (1,2,3):
    try:
        d=json.load(open("gpurun_out/adapt_pmc%d.json"%i))
    except Exception as e:
        print(i, e); continue
    for k,v in d.items():
        if "k_adapt_chol_apply" in k or "k_mh_steps" in k: print(k, v)
PY
