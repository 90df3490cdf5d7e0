#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/pipe -o t -- python $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > $O/pipe.log 2>&1
python $R/tools/trace_pipelined.py $O/pipe > $O/pipelined_trace.txt 2>&1
head -50 $O/pipelined_trace.txt
