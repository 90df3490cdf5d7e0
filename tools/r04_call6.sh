#!/bin/bash
# Round-4 GPU call 6: the split tier as two kernels (E) against one (D): 24-bit frames of 12 taps, config 4 itself, config 3; parity.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c6; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "lean24 or config4 or composed or stream" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s ms/step %.4f (min %.4f max %.4f) frac %.4f  alone %s" % (sys.argv[2], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["roofline"]["frac"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items() if v > 0.02}))
except Exception as e: print("ERR", sys.argv[2], e)
PY
}
for r in 1 2; do
 for v in D E; do
  export CLAXON_HIP_LIB=$PWD/claxon_amd/libclaxon_hip_$v.so
  timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config4 --order 12 --steps 48 > $O/$v.c4o12.$r.json 2> $O/$v.c4o12.$r.err; line $O/$v.c4o12.$r.json "$v config4 order 12 r$r"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config4 --steps 48 > $O/$v.c4.$r.json 2> $O/$v.c4.$r.err; line $O/$v.c4.$r.json "$v config4 r$r"
 done
done
unset CLAXON_HIP_LIB
python tools/stream_probe.py 10000 0 2>/dev/null | tail -1
python tools/stream_probe.py 10000 3334 2>/dev/null | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 > $O/c5.json 2> $O/c5.err; line $O/c5.json "config5 10k (compose auto)"
