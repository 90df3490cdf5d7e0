#!/bin/bash
# round 3: SQ counters of the fused lane kernels (lean vs general), one step at a time
set -u
cd "$(dirname "$0")/.."
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for P in lanes-fused lanes-general; do
  OUT=$R/gpurun_out/r03b_$P; mkdir -p $OUT
  BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline --path $P"
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH > "$OUT/pmc$i.log" 2>&1
  done
  python $R/tools/summarize_prof.py "$OUT" > $OUT/summary.txt 2>&1
  grep "clx_k_lean\|clx_k_lanes \|clx_k_scan" $OUT/summary.txt
done
