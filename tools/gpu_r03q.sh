#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03q; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s ms/step %.4f value %.0f frac %.4f issue %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"].get("issue",{}).get("frac")))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1].replace(".json",".err")).read()[-300:])
PY
}
for i in 1 2; do run c3_20_$i --steps 20 --warmup 5; done
run c3_96 --steps 96 --warmup 8
run c5_48 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48
run c5_share --workload config5 --shard-of 8 --shard-rank 3 --steps 24
run c2_20 --workload config2 --steps 20
run c4_20 --workload config4 --steps 20
for u in valu_cost vgpr_bank; do
  hipcc -O3 --offload-arch=gfx950 -o /tmp/$u tools/ubench/$u.hip 2> $O/$u.build.err && timeout 120 /tmp/$u > $O/ubench_$u.txt 2>&1
  tail -3 $O/ubench_$u.txt
done
bash tools/profile_r03.sh config5 2>&1 | tail -12
