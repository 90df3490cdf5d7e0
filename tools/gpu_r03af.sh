#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
O=gpurun_out/r03af; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/driver_line.json 2> $O/driver_line.err
tail -1 $O/driver_line.json | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('driver-style: value %.0f ms/step %.4f frac %.4f issue %.3f' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('issue',{}).get('frac',0)))
print(' merged_launch', j['roofline'].get('merged_launch'))
print(' cpu_baseline', j.get('cpu_baseline'))
"
grep real $O/driver_line.err
timeout 900 python tools/stress_gpu.py 3000 2 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_r03_config3; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pipe" -o t -- python $REPO/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > "$OUT/pipe.log" 2>&1
python $REPO/tools/trace_pipelined.py "$OUT/pipe" > "$OUT/pipelined_trace.txt" 2>&1
head -12 $OUT/pipelined_trace.txt
find $OUT/pipe -name "*kernel_stats.csv" | head -2
