#!/bin/bash
# A/B of two (or more) builds of the library on one box: tools/gpu_ab.sh "A B" rounds -- bench lines alternate between the builds
# (a build X is claxon_amd/libclaxon_hip_X.so:  CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_X.so CLX_EXTRA_FLAGS="-D..." python -c
#  "import claxon_amd as cx; cx.build(force=True)"  -- built .so files travel to the GPU box with the snapshot; box-to-box noise is
#  +-5 % on pipelined steps, so only builds that alternate on ONE box are compared)
set -u
cd "$(dirname "$0")/.."
VARS=${1:-"A B"}; ROUNDS=${2:-3}
O=gpurun_out/ab; mkdir -p $O
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_$v.so
    for cfg in "c3 --steps 48 --warmup 6" "c5 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48" "c4 --workload config4 --steps 48"; do
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
