#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03ah; mkdir -p $O
for r in 1 2 3; do for v in A B; do
  export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_$v.so
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 96 > $O/$v.$r.json 2> $O/$v.$r.err
  python - "$O/$v.$r.json" "$v c3 r$r" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m=j["roofline"].get("merged_launch",{})
    print("%-12s ms/step %.4f  alone %s  merged %s" % (sys.argv[2], j["ms_per_step"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items() if v > 0.05}, {k: round(v,3) for k,v in m.get("kernel_ms",{}).items() if v > 0.05}))
except Exception as e: print("ERR", sys.argv[2], e)
PY
done; done
