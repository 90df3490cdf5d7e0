#!/bin/bash
# Round-4 GPU call 3: waves composed by content -- the parity suite, config 5 (10 000 of 80 000 tiled frames and the 125 003-frame rank
# share of the 1 M-frame job) with clx_k_compose on / off, config 3 forced on (what the extra kernel costs where it buys nothing).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c3; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s ms/step %.4f (min %.4f max %.4f) frac %.4f  alone %s" % (sys.argv[2], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["roofline"]["frac"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items() if v > 0.004}))
except Exception as e: print("ERR", sys.argv[2], e)
PY
}
for r in 1 2; do
 for c in off on; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 --compose $c > $O/c5_10k_$c.$r.json 2> $O/c5_10k_$c.$r.err
  line $O/c5_10k_$c.$r.json "config5 10k compose=$c r$r"
 done
done
for c in off on; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --workload config5 --shard-of 8 --shard-rank 3 --steps 24 --compose $c > $O/c5_share_$c.json 2> $O/c5_share_$c.err
  line $O/c5_share_$c.json "config5 125003-frame share compose=$c"
done
for r in 1 2; do
 for c in off on; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 48 --compose $c > $O/c3_$c.$r.json 2> $O/c3_$c.$r.err
  line $O/c3_$c.$r.json "config3 compose=$c r$r"
 done
done
