#!/bin/bash
# the round's last check: GPU suite, smoke, the driver's bench command, and the default kernel choice for a small pipelined batch
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
tail -1 $O/driver_line.json | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('driver-style: value %.0f ms/step %.4f frac %.4f issue %.3f traffic %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('issue',{}).get('frac',0), j['roofline']['traffic']))"
python bench.py --no-cpu-baseline --no-extras --frames 2500 --steps 96 > $O/small.json 2> $O/small.err
tail -1 $O/small.json | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('2500 frames, auto: value %.0f ms/step %.4f kernels %s' % (j['value'], j['ms_per_step'], list(j['roofline']['kernel_ms'])))"
