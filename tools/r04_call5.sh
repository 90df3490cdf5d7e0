#!/bin/bash
# Round-4 GPU call 5: the host-to-host pipeline with tapering chunks against thirds; clx_k_compose with its loads in flight.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c5; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests -m gpu -x -q -k "stream or composed" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do
python tools/stream_probe.py 10000 0 2>/dev/null | tail -1
python tools/stream_probe.py 10000 3334 2>/dev/null | tail -1
done
python tools/stream_probe.py 20000 0 2>/dev/null | tail -1
python tools/stream_probe.py 20000 6667 2>/dev/null | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 --compose on > $O/c5.json 2> $O/c5.err
python - $O/c5.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config5 10k compose on: ms/step %.4f" % j["ms_per_step"], {k: round(v,3) for k,v in j["roofline"]["kernel_ms"].items()})
PY
