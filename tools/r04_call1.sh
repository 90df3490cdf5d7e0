#!/bin/bash
# Round-4 GPU call 1: the parity suite on the new build, A/B of the round-3 build (A) against the CRC-in-decode build (B) on one
# box, the default bench line, and the two-ranks-on-one-GPU rehearsal of the N-rank line.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/gpu_ab.sh "A B" 2 2>&1 | tee $O/ab.log
unset CLAXON_HIP_LIB
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 600 python bench.py --gpus 2 --devices 0,0 --backend gloo --steps 20 --warmup 5 --no-extras > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err
echo "2-rank rc=$?"; tail -c 400 $O/bench_2ranks_1gpu.json; tail -5 $O/bench_2ranks_1gpu.err
