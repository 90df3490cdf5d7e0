#!/bin/bash
# Round-4 GPU call 11: K (committed) against L (ballot votes, calm waves take their second granule together).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c11; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "fused or composed or scale or oracle" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "K L" 2 2>&1 | tee $O/ab.log
