#!/bin/bash
# Round-4 GPU call 10: H (first dot as VOP3P) against K (+ calm waves pump their rings every other turn -- decided once per wave).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c10; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "fused or composed or scale or oracle" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "H K" 2 2>&1 | tee $O/ab.log
unset CLAXON_HIP_LIB
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_20_$i.json 2> $O/bench_20_$i.err
python - $O/bench_20_$i.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps20: ms/step median %.4f min %.4f max %.4f  value %.0f" % (j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["value"]), j["roofline"].get("merged_launch",{}).get("kernel_ms"))
PY
done
