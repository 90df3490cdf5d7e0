#!/bin/bash
# Round-4 GPU call 16: K (committed) against P (calm waves ask for their second and third granule at every fourth pump only; calm = at
# most 6 bits per sample): configs 3 and 2 (calm), 5 (never calm), parity subset of P.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "fused or composed or oracle" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for r in 1 2 3; do
 for v in K P; do
  export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_$v.so
  for cfg in "c3 --steps 48" "c2 --workload config2 --steps 48" "c5 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48"; do
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
