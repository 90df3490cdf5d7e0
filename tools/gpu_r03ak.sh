#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/profile_r03.sh config2 2>&1 | grep -v "^clx_k\|^$" | tail -4
O=gpurun_out/r03ak; mkdir -p $O
for w in config2 config4; do
python bench.py --no-cpu-baseline --no-extras --workload $w --steps 20 --warmup 5 > $O/$w.json 2> $O/$w.err; tail -1 $O/$w.json | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w value %.0f ms/step %.4f frac %.4f' % (j['value'], j['ms_per_step'], j['roofline']['frac']))"
done
