#!/bin/bash
# Round-4 GPU call 2: parity of the pair-form build, A/B of B (CRC in the decode lanes) against C (+ pair form, per-launch events),
# the driver-style bench line.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c2; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/gpu_ab.sh "B C" 2 2>&1 | tee $O/ab.log
unset CLAXON_HIP_LIB
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_20_$i.json 2> $O/bench_20_$i.err
python - $O/bench_20_$i.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps20: ms/step median %.4f min %.4f max %.4f  value %.0f" % (j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["value"]), j["roofline"].get("merged_launch",{}).get("kernel_ms"))
PY
done
