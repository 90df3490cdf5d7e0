#!/bin/bash
# Round-4 GPU call 14: K (committed) against N (the scan's codes two by two out of one window register).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c14; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "fused or composed or scale or oracle" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "K N" 2 2>&1 | tee $O/ab.log
