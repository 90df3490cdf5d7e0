#!/bin/bash
# Round-4 GPU call 9: H (first dot as VOP3P) against J (+ ring pumped every other turn unless hungry, the question asked every eighth
# turn while hungry); parity subset of J.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c9; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "fused or composed or scale or oracle" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "H J" 3 2>&1 | tee $O/ab.log
