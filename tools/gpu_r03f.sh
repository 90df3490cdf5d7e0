#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 1 2 3 6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$M -o t -- python $R/tools/merge_probe.py $M 5 > $O/m$M.log 2>&1
  echo "== M=$M"; tail -2 $O/m$M.log; grep -h "clx_k_lean\|clx_k_scan\|clx_k_crc16" $O/m$M/*kernel_stats.csv | cut -d, -f1-4
done
M=6
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc1 -o p -- python $R/tools/merge_probe.py $M 3 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -o p -- python $R/tools/merge_probe.py $M 3 > $O/pmc2.log 2>&1
python $R/tools/summarize_prof.py $O > $O/summary.txt 2>&1
grep "clx_k_lean .*SQ_\|clx_k_lean .*GRBM\|clx_k_scan .*SQ_LDS" $O/summary.txt
