#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r03g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes-fused" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for M in 1 3 6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$M -o t -- python $R/tools/merge_probe.py $M 5 > $O/m$M.log 2>&1
  echo "== M=$M"; grep "exact" $O/m$M.log; grep -h "clx_k_lean\|clx_k_scan\|clx_k_crc16" $O/m$M/*kernel_stats.csv | cut -d, -f1-4
done
cd $R
timeout 600 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras > $O/bench20.json 2> $O/bench20.err
timeout 600 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-extras > $O/bench96.json 2> $O/bench96.err
timeout 600 python bench.py --workload config2 --frames 20000 --steps 96 --no-cpu-baseline --no-extras --path lanes-fused > $O/bench_c2.json 2> $O/bench_c2.err
for f in $O/bench*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], j["roofline"]["kernel_ms"])
except Exception as e: print("  ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
PY
done
