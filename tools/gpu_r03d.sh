#!/bin/bash
# round 3: merged launches of the fused lane kernels -- whole GPU suite, then the bench workload
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
for Q in 4 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras > $O/bench20_q$Q.json 2> $O/bench20_q$Q.err
  GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-extras > $O/bench96_q$Q.json 2> $O/bench96_q$Q.err
done
GPU_MAX_HW_QUEUES=4 timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-extras --path lanes-general > $O/bench96_general.json 2> $O/bench96_general.err
GPU_MAX_HW_QUEUES=4 timeout 600 python bench.py --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 --no-cpu-baseline --no-extras > $O/bench_c5.json 2> $O/bench_c5.err
GPU_MAX_HW_QUEUES=4 timeout 600 python bench.py --workload config2 --frames 20000 --steps 96 --no-cpu-baseline --no-extras --path lanes-fused > $O/bench_c2.json 2> $O/bench_c2.err
python tools/submit_probe.py 10000 > $O/submit_probe.txt 2>&1
for f in $O/bench*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "inflight", j["config"]["steps_in_flight"])
except Exception as e: print("  ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
PY
done
tail -6 $O/submit_probe.txt
