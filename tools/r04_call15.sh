#!/bin/bash
# Round-4 GPU call 15: a burst's first launch made small when the machine is idle (CLX_TUNE_WARM in the tuning build: 0 = off): the
# 20-step region from idle, the 96-step one, the driver-style bench line; pipeline tests of the product build (warm 3).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c15; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_T.so
for wm in 0 2 3 4 6 0 3; do
  echo "== warm $wm"; CLX_TUNE_WARM=$wm python tools/region_probe.py 20 5 2>/dev/null | awk '{print $(NF-2)}' | tr '\n' ' '; echo
done
for wm in 0 3; do
  CLX_TUNE_WARM=$wm timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b20_$wm.json 2> $O/b20_$wm.err
  CLX_TUNE_WARM=$wm timeout 300 python bench.py --steps 96 --no-cpu-baseline --no-extras > $O/b96_$wm.json 2> $O/b96_$wm.err
  python - $O/b20_$wm.json $O/b96_$wm.json $wm <<'PY'
import json,sys
a=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("warm %s: 20 steps %.4f (min %.4f max %.4f) | 96 steps %.4f (min %.4f max %.4f)" % (sys.argv[3], a["ms_per_step"], a["ms_per_step_min"], a["ms_per_step_max"], b["ms_per_step"], b["ms_per_step_min"], b["ms_per_step_max"]))
PY
done
