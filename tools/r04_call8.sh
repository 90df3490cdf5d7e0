#!/bin/bash
# Round-4 GPU call 8: F (committed) | H (+ first dot as VOP3P with a constant addend) | I (+ ring pumped every other turn unless a
# lane is hungry): parity of I, then the three workloads alternating; driver-style bench of I.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c8; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "F H I" 2 2>&1 | tee $O/ab.log
unset CLAXON_HIP_LIB
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_20.json 2> $O/bench_20.err
python - $O/bench_20.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps20: ms/step median %.4f min %.4f max %.4f  value %.0f" % (j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["value"]), j["roofline"].get("merged_launch",{}).get("kernel_ms"))
PY
