#!/bin/bash
# Round-4 GPU call 13: K (committed) against M (a merged launch's tail -- general kernels, results, stand-alone CRC -- on a stream of
# its own; three sets of scratch): pipeline / scale tests of M, the three workloads alternating, the driver-style line, a kernel
# trace of M's pipelined steps.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c13; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_scale.py tests/test_gpu_multictx.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/gpu_ab.sh "K M" 2 2>&1 | tee $O/ab.log
unset CLAXON_HIP_LIB
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_20.json 2> $O/bench_20.err
python - $O/bench_20.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps20: ms/step median %.4f min %.4f max %.4f  value %.0f  in flight %s launches %s" % (j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["value"], j["config"]["steps_in_flight"], j["config"]["merged_launches_per_region"]))
PY
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/pipe -o t -- python $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-extras > $R/$O/pipe.log 2>&1
python $R/tools/trace_pipelined.py $R/$O/pipe > $R/$O/pipelined_trace.txt 2>&1; head -12 $R/$O/pipelined_trace.txt
