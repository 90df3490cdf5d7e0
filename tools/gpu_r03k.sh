#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03k; mkdir -p $O
export CLAXON_HIP_LIB=$R/claxon_amd/libclaxon_hip_r24.so
export CLX_TUNE_MERGE=12 CLX_TUNE_STREAMS=1
cd /tmp && export TMPDIR=/tmp
for M in 9 12; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$M -o t -- python $R/tools/merge_probe.py $M 4 > $O/m$M.log 2>&1
  echo "== M=$M"; tail -2 $O/m$M.log; grep -h "clx_k_lean\|clx_k_scan\|clx_k_crc16" $O/m$M/*kernel_stats.csv | cut -d, -f1-4
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_LEVEL_WAVES --kernel-trace --output-format csv -d $O/pmc12 -o p -- python $R/tools/merge_probe.py 12 3 > $O/pmc12.log 2>&1
python $R/tools/summarize_prof.py $O/pmc12 > $O/summary12.txt 2>&1
grep "clx_k_lean .*SQ\|clx_k_lean .*GRBM\|clx_k_scan .*SQ\|clx_k_scan .*GRBM" $O/summary12.txt
