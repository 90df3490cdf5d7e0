#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03c; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/pipe -o t -- python $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > $O/pipe.log 2>&1
python $R/tools/trace_pipelined.py $O/pipe > $O/pipelined_trace.txt 2>&1
head -60 $O/pipelined_trace.txt
# the same with HIP's default number of hardware queues
unset GPU_MAX_HW_QUEUES
GPU_MAX_HW_QUEUES=4 timeout 300 python $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > $O/bench_q4.json 2> $O/bench_q4.err
GPU_MAX_HW_QUEUES=16 timeout 300 python $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > $O/bench_q16.json 2> $O/bench_q16.err
# mono (no scan): config2 pipelined, lean vs general
GPU_MAX_HW_QUEUES=16 timeout 300 python $R/bench.py --workload config2 --frames 20000 --steps 48 --warmup 12 --no-cpu-baseline --no-extras --path lanes-fused > $O/bench_c2_lean.json 2> $O/bench_c2_lean.err
GPU_MAX_HW_QUEUES=16 timeout 300 python $R/bench.py --workload config2 --frames 20000 --steps 48 --warmup 12 --no-cpu-baseline --no-extras --path lanes-general > $O/bench_c2_general.json 2> $O/bench_c2_general.err
for f in $O/bench*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "inflight", j["config"]["steps_in_flight"])
except Exception as e: print("  ERR", e)
PY
done
