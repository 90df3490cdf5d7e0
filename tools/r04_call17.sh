#!/bin/bash
# Round-4 GPU call 17: P (committed) against Q (waves that pump every turn take their second granule at every other turn, more only
# when a lane is a turn's worth behind): config 5 composed and in stream order, config 4, config 3.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c17; mkdir -p $O
for r in 1 2; do
 for v in P Q; do
  export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_$v.so
  for cfg in "c5 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48" "c5s --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 --compose off" "c4 --workload config4 --steps 48" "c3 --steps 48"; do
    set -- $cfg; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$v.$name.$r.json 2> $O/$v.$name.$r.err
    python - "$O/$v.$name.$r.json" "$v $name r$r" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s ms/step %.4f  alone %s" % (sys.argv[2], j["ms_per_step"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items() if v > 0.05}))
except Exception as e: print("ERR", sys.argv[2], e)
PY
  done
 done
done
