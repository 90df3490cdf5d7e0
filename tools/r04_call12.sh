#!/bin/bash
# Round-4 GPU call 12: runs per merged launch x internal streams at the driver's 20 steps and at 96 (tuning build: -DCLX_TUNING);
# one batch at a time at four sizes (does the residual round trip of the wave kernels matter? 164 MB of output fit the 256 MB
# Infinity Cache, 328 MB do not).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c12; mkdir -p $O
export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_T.so
bash tools/merge_sweep.sh "12 2" "10 2" "8 2" "8 3" "6 4" 2>&1 | tee $O/merge_sweep.log
unset CLAXON_HIP_LIB
for n in 2500 5000 10000 20000; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-pipeline --path waves --frames $n --steps 20 > $O/one_$n.json 2> $O/one_$n.err
  python - $O/one_$n.json $n <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); n=int(sys.argv[2])
k=j["roofline"]["kernel_ms"]
print("one batch of %5d frames: ms/step %.4f  per 10k frames %.4f  kernels %s" % (n, j["ms_per_step"], j["ms_per_step"]*10000/n, {a: round(b,4) for a,b in k.items()}))
PY
done
