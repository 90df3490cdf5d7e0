#!/bin/bash
# round 3, first GPU call: parity of the lane builds (with / without the lean 16-bit tier), then A/B of the bench workload
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03a; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes-fused or lanes-general" > $O/pytest_lanes.log 2>&1; echo "pytest rc=$?" >> $O/pytest_lanes.log
tail -3 $O/pytest_lanes.log
for P in auto lanes-general; do
  timeout 600 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras --path $P > $O/bench20_$P.json 2> $O/bench20_$P.err
  timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-extras --path $P > $O/bench96_$P.json 2> $O/bench96_$P.err
done
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-pipeline --path lanes-fused > $O/bench_np_lean.json 2> $O/bench_np_lean.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-pipeline --path lanes-general > $O/bench_np_general.json 2> $O/bench_np_general.err
timeout 600 python bench.py --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 20 --no-cpu-baseline --no-extras > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python bench.py --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 20 --no-cpu-baseline --no-extras --path lanes-general > $O/bench_c5_general.json 2> $O/bench_c5_general.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_np -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline --path lanes-fused > $R/$O/trace_np.log 2>&1
cd $R
for f in $O/bench*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "kernels", j["roofline"]["kernel_ms"])
except Exception as e: print("  ERR", e)
PY
done
find $O/trace_np -name "*kernel_stats.csv" | head -1 | xargs -r head -12
