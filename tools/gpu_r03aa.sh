#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03aa; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s ms/step %.4f value %.0f frac %.4f %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["roofline"]["frac"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items()}))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run c3_20 --steps 20 --warmup 5
run c3_20b --steps 20 --warmup 5
run c3_96 --steps 96 --warmup 5
run c5_48 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48
run c5_share --workload config5 --shard-of 8 --shard-rank 3 --steps 24
run c2_48 --workload config2 --steps 48
run c4_48 --workload config4 --steps 48
