#!/bin/bash
# Round-2 rocprofv3 evidence (run through gpurun): for each BASELINE shape, a kernel trace with stats and the two HBM counters in
# separate --pmc passes of `bench.py --no-pipeline` (steps one at a time, so that every kernel's duration is its own); for the
# bench workload also the SQ instruction / wait counters, and a kernel trace of the pipelined steps (how far the predictor stage
# of one step overlaps the Rice stage of the next).  Outputs under gpurun_out/prof_r02_<name>/; summaries go to profiles/.
# usage: tools/profile_r02.sh [names...]      names: config3 config2 config4 config5 (default: all)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAMES=${*:-config3 config2 config4 config5}
cd /tmp && export TMPDIR=/tmp
for NAME in $NAMES; do
  case $NAME in
    config5) ARGS="--workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3";;
    config3_lanes) ARGS="--workload config3 --frames 10000 --path lanes-fused";;      # the fused lane kernels (what pipelined submissions run)
    *)       ARGS="--workload $NAME --frames 10000";;
  esac
  OUT=$REPO/gpurun_out/prof_r02_$NAME
  mkdir -p "$OUT"
  BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline $ARGS"
  echo "== $NAME: kernel trace"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
  grep '"metric"' "$OUT/trace.log" | tail -1 > "$OUT/bench_line.json"
  i=0
  SETS=("FETCH_SIZE" "WRITE_SIZE")
  if [ "$NAME" = config3 ]; then
    SETS+=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
  fi
  for set in "${SETS[@]}"; do
    i=$((i+1))
    echo "== $NAME: pmc $i: $set"
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH > "$OUT/pmc$i.log" 2>&1
  done
  python $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
  if [ "$NAME" = config3 ]; then
    echo "== config3: kernel trace of pipelined steps"
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pipe" -o t -- python $REPO/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/pipe.log" 2>&1
    python $REPO/tools/trace_overlap.py "$OUT/pipe" > "$OUT/pipeline_overlap.txt" 2>&1
  fi
  tail -5 "$OUT/summary.txt"
done
