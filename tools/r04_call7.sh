#!/bin/bash
# Round-4 GPU call 7: F (committed) against G (first dot as VOP3P with a constant addend, no clamp behind v_ffbh): parity, then the
# usual three workloads alternating.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c7; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "F G" 2 2>&1 | tee $O/ab.log
