#!/bin/bash
# bench.py under CLX_TUNE_MERGE / CLX_TUNE_STREAMS: runs per merged launch of the fused lane kernels x internal streams (run through
# gpurun; the numbers are collected in profiles/r03_merge_sweep.txt).  The product build reads no environment variable: point
# CLAXON_HIP_LIB at a build made with CLX_EXTRA_FLAGS="-DCLX_TUNING" (tools/r04_call12.sh shows how).
# usage: tools/merge_sweep.sh ["M S" ...] [-- bench.py arguments]      default: a few shapes, the bench workload, 20 and 96 steps
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/merge_sweep; mkdir -p $O
SHAPES=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; SHAPES+=("$1"); shift; done
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("12 2" "6 2" "6 4" "4 3" "12 1")
for cfg in "${SHAPES[@]}"; do
  set -- $cfg
  for steps in 20 96; do
    CLX_TUNE_MERGE=$1 CLX_TUNE_STREAMS=$2 timeout 300 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-extras "${EXTRA[@]}" > $O/b_m$1_s$2_$steps.json 2> $O/b_m$1_s$2_$steps.err
    python - "$O/b_m$1_s$2_$steps.json" "$1" "$2" "$steps" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("merge %s streams %s steps %s: ms/step %.4f value %.0f in flight %s" % (sys.argv[2], sys.argv[3], sys.argv[4], j["ms_per_step"], j["value"], j["config"].get("steps_in_flight")))
except Exception as e: print("ERR", sys.argv[1], e)
PY
  done
done
