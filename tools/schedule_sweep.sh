#!/bin/bash
# The schedule of the merged launches, measured (round 6): runs in the first launch behind a wait (CLX_TUNE_FIRST) and priorities of the two
# internal streams (CLX_TUNE_PRIO) -- read by builds made with -DCLX_TUNING only (CLAXON_HIP_LIB points at one).
# usage: tools/schedule_sweep.sh rounds "variants" "cfgs"      variant = first:prio[:merge[:streams]], 0 = the default
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
O=$REPO/gpurun_out/r06; mkdir -p $O
ROUNDS=${1:-2}; VARS=${2:-"0:0 6:0 0:1"}; CFGS=${3:-"c3d c3l"}
export CLAXON_HIP_LIB=${CLAXON_HIP_LIB:-$REPO/claxon_amd/libclaxon_hip_tune.so}
cd $REPO
for r in $(seq 1 $ROUNDS); do for v in $VARS; do
  IFS=: read first prio merge streams <<< "$v"
  unset CLX_TUNE_FIRST CLX_TUNE_PRIO CLX_TUNE_MERGE CLX_TUNE_STREAMS
  [ "${first:-0}" != 0 ] && export CLX_TUNE_FIRST=$first
  [ "${prio:-0}" != 0 ] && export CLX_TUNE_PRIO=$prio
  [ -n "${merge:-}" ] && [ "$merge" != 0 ] && export CLX_TUNE_MERGE=$merge
  [ -n "${streams:-}" ] && [ "$streams" != 0 ] && export CLX_TUNE_STREAMS=$streams
  for name in $CFGS; do
    case $name in
      c3)  args="--steps 48 --warmup 6" ;;
      c3d) args="--steps 20 --warmup 5" ;;
      c3l) args="--steps 96 --warmup 8" ;;
      c2)  args="--workload config2 --steps 48" ;;
      c5)  args="--workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48" ;;
      c4)  args="--workload config4 --steps 48" ;;
    esac
    f=$O/sched_$v.$name.$r
    timeout 300 python bench.py --no-cpu-baseline --no-extras $args > "$f.json" 2> "$f.err"
    python - "$f.json" "$v $name r$r" <<'PY' | tee -a $O/sched.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s ms/step %.4f (min %.4f max %.4f) in flight %s" % (sys.argv[2], j["ms_per_step"], j.get("ms_per_step_min", 0), j.get("ms_per_step_max", 0), j["config"].get("steps_in_flight")))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1][:-5]+".err").read()[-600:])
PY
  done
done; done
