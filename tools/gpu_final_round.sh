#!/bin/bash
# A round's figures of record on one box (ROUND=r06 by default: the output directory's suffix): the GPU suite, smoke(), the driver-style bench line (twice), the default line with its
# secondary figures, configs 2 / 4 / 5 pipelined, the 125 003-frame rank share of config 5, two ranks on one GPU (gloo).
set -u
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r06}
O=gpurun_out/final_$ROUND; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
[ -n "${SKIP_TESTS:-}" ] || { timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j["roofline"]
    print("%-30s ms/step %.4f (min %.4f max %.4f) value %.0f frac %.4f traffic %s valu/sample %s" % (sys.argv[2], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_max"], j["value"], r["frac"], r.get("traffic"), (r.get("issue") or {}).get("valu_per_sample")))
except Exception as e: print("ERR", sys.argv[2], e)
PY
}
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; line $O/bench_driver_$i.json "driver-style $i"
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; line $O/bench_default.json "default (96 steps)"
timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config2 --steps 48 > $O/c2.json 2> $O/c2.err; line $O/c2.json "config2"
timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config4 --steps 48 > $O/c4.json 2> $O/c4.err; line $O/c4.json "config4"
timeout 300 python bench.py --no-cpu-baseline --no-extras --workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48 > $O/c5_10k.json 2> $O/c5_10k.err; line $O/c5_10k.json "config5 10k"
timeout 600 python bench.py --no-cpu-baseline --no-extras --workload config5 --shard-of 8 --shard-rank 3 --steps 24 > $O/c5_share.json 2> $O/c5_share.err; line $O/c5_share.json "config5 125003-frame share"
timeout 600 python bench.py --gpus 2 --devices 0,0 --backend gloo --steps 20 --warmup 5 --no-extras > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err; echo "2-rank rc=$?"; line $O/bench_2ranks_1gpu.json "2 ranks on one GPU (gloo)"
# the N-rank line's collectives on the hardware that exists: ONE rank, process group through RCCL (init_process_group("nccl", device_id=...), barrier,
# MAX / SUM all-reduces and the all-gather on device tensors)
MASTER_ADDR=127.0.0.1 timeout 600 python bench.py --gpus 1 --process-group --backend nccl --steps 20 --warmup 5 --no-extras > $O/bench_rccl_one_rank.json 2> $O/bench_rccl_one_rank.err; echo "rccl one rank rc=$?"; line $O/bench_rccl_one_rank.json "1 rank, RCCL process group"
python - $O/bench_rccl_one_rank.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   process_group:", j["config"]["process_group"])
except Exception as e: print("ERR", e)
PY

