#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03ad; mkdir -p $O
export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_B.so
for cfg in "16 2" "12 2" "8 4" "16 1"; do set -- $cfg
  for steps in 20 96; do
    CLX_TUNE_MERGE=$1 CLX_TUNE_STREAMS=$2 timeout 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-extras > $O/b_m$1_s$2_$steps.json 2> $O/b_m$1_s$2_$steps.err
    python - "$O/b_m$1_s$2_$steps.json" "$1" "$2" "$steps" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("merge %s streams %s steps %s: ms/step %.4f value %.0f in flight %s" % (sys.argv[2], sys.argv[3], sys.argv[4], j["ms_per_step"], j["value"], j["config"].get("steps_in_flight")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
  done
done
