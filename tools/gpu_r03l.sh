#!/bin/bash
# round 3: whole GPU suite, then the bench workloads (config 3 default, 2, 4, 5) and the general lane kernels for comparison
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-18s ms/step %.4f value %.0f frac %.4f inflight %s %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["roofline"]["frac"], j["config"]["steps_in_flight"], {k: round(v,3) for k,v in j["roofline"]["kernel_ms"].items()}))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
run c3_20 --steps 20 --warmup 8
run c3_96 --steps 96 --warmup 8
run c3_96_general --steps 96 --warmup 8 --path lanes-general
run c3_one --steps 20 --no-pipeline
run c2_96 --workload config2 --steps 96
run c4_48 --workload config4 --steps 48
run c4_48_np --workload config4 --steps 20 --no-pipeline
run c5_48 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48
run c5_48_general --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 --path lanes-general
