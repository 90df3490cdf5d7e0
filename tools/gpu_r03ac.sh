#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03ac; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s ms/step %.4f value %.0f frac %.4f %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["roofline"]["frac"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items()}))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
run c3_20 --steps 20 --warmup 5
run c5_share --workload config5 --shard-of 8 --shard-rank 3 --steps 24
run c3_np --steps 10 --no-pipeline
bash tools/profile_r03.sh config3 config5 config4 2>&1 | grep -v "^clx_k\|^$" | tail -30
