#!/bin/bash
# pipelined submissions of small batches: the wave kernels (four in flight) against the fused lane kernels (merged launches) --
# the measurement behind clx_select_path's `pipelined` thresholds (clx_plan.h).  Run through gpurun.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/path_sweep; mkdir -p $O
for w in config3 config2; do for n in 600 1250 2500 5000; do for path in waves lanes-fused; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --workload $w --frames $n --path $path --steps 96 > $O/$w.$n.$path.json 2> $O/$w.$n.$path.err
  python - "$O/$w.$n.$path.json" "$w frames $n $path" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s ms/step %.4f  Gsamples/s %.1f  in flight %s" % (sys.argv[2], j["ms_per_step"], j["value"]/1e3, j["config"].get("steps_in_flight")))
except Exception as e: print("ERR", sys.argv[2], e)
PY
done; done; done
