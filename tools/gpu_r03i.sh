#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03i; mkdir -p $O
for V in r24 r32 r24 r32; do
  if [ $V = r24 ]; then export CLAXON_HIP_LIB=$R/claxon_amd/libclaxon_hip_r24.so; else unset CLAXON_HIP_LIB; fi
  timeout 300 python bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-extras > $O/b96_$V.json 2> $O/b96_$V.err
  timeout 300 python tools/merge_probe.py 6 6 > $O/m6_$V.log 2>&1
  python - "$O/b96_$V.json" $V <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "96 steps: ms/step %.4f" % j["ms_per_step"], j["roofline"]["kernel_ms"])
PY
  tail -3 $O/m6_$V.log
done
