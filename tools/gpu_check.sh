#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/check; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/stress_gpu.py 5000 1 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
