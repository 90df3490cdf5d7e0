#!/bin/bash
# A round's rocprofv3 evidence (run through gpurun; ROUND=r06 by default: the outputs' prefix).  For the bench workload (config 3, the kernels of the TIMED steps: fused lane
# kernels with the lean 16-bit tier): kernel trace + stats of steps one at a time, of ONE merged launch of twelve steps, and of the
# pipelined steps (the timed mode); HBM counters and SQ instruction / wait counters in separate --pmc passes.  For configs 2 / 4 / 5:
# kernel stats and HBM counters of steps one at a time.  Outputs under gpurun_out/prof_${ROUND}_<name>/; summaries go to profiles/.
# usage: [ROUND=r06] tools/profile_round.sh [names...]     names: config3 config2 config4 config5 config5share (default: the first four; config5share: one rank's share of the 1 M-frame job, 125 003 frames)
set -u
ROUND=${ROUND:-r06}
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAMES=${*:-config3 config2 config4 config5}
cd /tmp && export TMPDIR=/tmp
for NAME in $NAMES; do
  case $NAME in
    config5) ARGS="--workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --path lanes-fused";;
    config5share) ARGS="--workload config5 --shard-of 8 --shard-rank 3 --path lanes-fused";;
    config4) ARGS="--workload config4 --frames 10000 --path lanes-fused";;
    *)       ARGS="--workload $NAME --frames 10000 --path lanes-fused";;
  esac
  OUT=$REPO/gpurun_out/prof_${ROUND}_$NAME
  mkdir -p "$OUT"
  BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline $ARGS"
  echo "== $NAME: kernel trace (one step at a time)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
  grep '"metric"' "$OUT/trace.log" | tail -1 > "$OUT/bench_line.json"
  SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    echo "== $NAME: pmc $i: $set"
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH > "$OUT/pmc$i.log" 2>&1
  done
  python $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
  if [ "$NAME" = config3 ]; then
    echo "== config3: one merged launch of twelve steps at a time"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/merged12" -o t -- python $REPO/tools/merge_probe.py 12 5 > "$OUT/merged12.log" 2>&1
    echo "== config3: kernel trace of the pipelined steps (the timed mode)"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pipe" -o t -- python $REPO/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > "$OUT/pipe.log" 2>&1
    python $REPO/tools/trace_pipelined.py "$OUT/pipe" > "$OUT/pipelined_trace.txt" 2>&1
    for set in "FETCH_SIZE" "WRITE_SIZE"; do
      timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/merged12_$set" -o p -- python $REPO/tools/merge_probe.py 12 3 > "$OUT/merged12_$set.log" 2>&1
    done
  fi
  tail -8 "$OUT/summary.txt"
done
