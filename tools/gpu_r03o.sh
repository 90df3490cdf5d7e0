#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03o; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s ms/step %.4f value %.0f frac %.4f issue %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"].get("issue",{}).get("frac")))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
for i in 1 2 3; do run c3_20_$i --steps 20 --warmup 5; done
run c3_96 --steps 96 --warmup 8
run c5_48 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48
