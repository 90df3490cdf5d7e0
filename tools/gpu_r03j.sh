#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03j; mkdir -p $O
export CLAXON_HIP_LIB=$R/claxon_amd/libclaxon_hip_r24.so
cd /tmp && export TMPDIR=/tmp
for M in 1 6; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/m${M}_a -o p -- python $R/tools/merge_probe.py $M 3 > $O/m${M}_a.log 2>&1
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $O/m${M}_b -o p -- python $R/tools/merge_probe.py $M 3 > $O/m${M}_b.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/m${M}_c -o p -- python $R/tools/merge_probe.py $M 3 > $O/m${M}_c.log 2>&1
  mkdir -p $O/m$M; cp -r $O/m${M}_a $O/m${M}_b $O/m${M}_c $O/m$M/ 2>/dev/null
  python $R/tools/summarize_prof.py $O/m$M > $O/summary_m$M.txt 2>&1
  echo "== M=$M"; grep "clx_k_lean .*SQ\|clx_k_lean .*GRBM" $O/summary_m$M.txt
done
