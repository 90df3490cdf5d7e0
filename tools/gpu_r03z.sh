#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03z; mkdir -p $O
export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_d24.so
run() { m=$1; s=$2; name=$3; shift 3
    CLX_TUNE_MERGE=$m CLX_TUNE_STREAMS=$s timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" > $O/b_${name}_m${m}_s${s}.json 2> $O/b_${name}_m${m}_s${s}.err
    python - "$O/b_${name}_m${m}_s${s}.json" "$name merge $m streams $s" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s: ms/step %.4f value %.0f" % (sys.argv[2], j["ms_per_step"], j["value"]))
except Exception as e: print("ERR", sys.argv[1], e)
PY
}
for cfg in "12 1" "12 2" "10 2" "9 2" "11 2" "6 4"; do set -- $cfg; run $1 $2 c3_20 --steps 20 --warmup 5; run $1 $2 c3_96 --steps 96 --warmup 5; done
for cfg in "6 2" "12 2"; do set -- $cfg
  run $1 $2 c5 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48
  run $1 $2 c4 --workload config4 --steps 48
  run $1 $2 c2 --workload config2 --steps 48
done
