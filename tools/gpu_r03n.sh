#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in lanes-fused lanes-general; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$P -o p -- python $R/bench.py --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pipeline --path $P > $O/$P.log 2>&1
  python $R/tools/summarize_prof.py $O/$P > $O/summary_$P.txt 2>&1
  echo "== $P"; grep "SQ_INSTS_VALU\|SQ_WAVE_CYCLES \|SQ_INSTS_SALU" $O/summary_$P.txt | grep "clx_k_lean\|clx_k_lanes \|clx_k_lanes_hi\|clx_k_scan"
done
