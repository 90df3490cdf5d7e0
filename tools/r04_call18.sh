#!/bin/bash
# Round-4 GPU call 18: P (committed) against R (the stand-alone CRC kernel behind merged lane launches: 256 workgroups per run, and a
# workgroup with nothing to do leaves before it copies its tables).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -k "crc or pipeline or submit" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for r in 1 2 3; do
 for v in P R; do
  export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_$v.so
  for cfg in "c3 --steps 48" "c3d --gpus 1 --steps 20 --warmup 5" "c5 --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48"; do
    set -- $cfg; name=$1; shift
    timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" > $O/$v.$name.$r.json 2> $O/$v.$name.$r.err
    python - "$O/$v.$name.$r.json" "$v $name r$r" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s ms/step %.4f (min %.4f)" % (sys.argv[2], j["ms_per_step"], j["ms_per_step_min"]))
except Exception as e: print("ERR", sys.argv[2], e)
PY
  done
 done
done
