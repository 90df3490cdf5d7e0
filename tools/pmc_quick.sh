#!/bin/bash
# One or two rocprofv3 --pmc passes over a short bench.py run (through gpurun): tools/pmc_quick.sh <tag> "<counters>" ["<counters>"]
set -u
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH > "$OUT/pmc$i.log" 2>&1
  tail -1 "$OUT/pmc$i.log" | cut -c1-200
done
python $REPO/tools/summarize_prof.py "$OUT" 2>&1 | grep -v "^$" | grep "clx_k_residual\|clx_k_predict" 
