#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03y; mkdir -p $O
export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_d24.so
for cfg in "6 2" "8 2" "8 1" "7 2" "5 2" "4 3" "8 3" "12 2" "4 4"; do
  set -- $cfg
  for steps in 20 96; do
    CLX_TUNE_MERGE=$1 CLX_TUNE_STREAMS=$2 timeout 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-extras > $O/b_m$1_s$2_$steps.json 2> $O/b_m$1_s$2_$steps.err
    python - "$O/b_m$1_s$2_$steps.json" "$1" "$2" "$steps" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("merge %s streams %s steps %s: ms/step %.4f value %.0f" % (sys.argv[2], sys.argv[3], sys.argv[4], j["ms_per_step"], j["value"]))
except Exception as e: print("ERR", sys.argv[1], e)
PY
  done
done
