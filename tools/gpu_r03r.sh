#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03r; mkdir -p $O
export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_d24.so
for cfg in "6 2 0" "6 3 0" "6 4 0" "8 2 0" "8 3 0" "9 2 0" "12 2 0" "4 6 16" "6 4 16" "8 3 16" "3 6 16"; do
  set -- $cfg
  if [ "$3" = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$3; fi
  for steps in 20 96; do
    CLX_TUNE_MERGE=$1 CLX_TUNE_STREAMS=$2 timeout 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-extras > $O/b_m$1_s$2_q$3_$steps.json 2> $O/b_m$1_s$2_q$3_$steps.err
    python - "$O/b_m$1_s$2_q$3_$steps.json" "$1" "$2" "$3" "$steps" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("merge %s streams %s queues %s steps %s: ms/step %.4f value %.0f" % (sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], j["ms_per_step"], j["value"]))
except Exception as e: print("ERR", sys.argv[1], e)
PY
  done
done
