#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03ab; mkdir -p $O
for cfg in "1 2" "2 2" "3 2" "4 2" "6 2" "2 3" "1 4"; do set -- $cfg
    CLX_TUNE_MERGE=$1 CLX_TUNE_STREAMS=$2 timeout 600 python bench.py --no-cpu-baseline --no-extras --workload config5 --shard-of 8 --shard-rank 3 --steps 24 > $O/share_m$1_s$2.json 2> $O/share_m$1_s$2.err
    python - "$O/share_m$1_s$2.json" "c5_share merge $1 streams $2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s: ms/step %.4f value %.0f" % (sys.argv[2], j["ms_per_step"], j["value"]))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
for cfg in "12 2" "8 2" "6 2" "4 2"; do set -- $cfg
    CLX_TUNE_MERGE=$1 CLX_TUNE_STREAMS=$2 timeout 600 python bench.py --no-cpu-baseline --no-extras --frames 30000 --steps 48 > $O/c3_30k_m$1_s$2.json 2> $O/c3_30k_m$1_s$2.err
    python - "$O/c3_30k_m$1_s$2.json" "c3 30000 frames merge $1 streams $2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s: ms/step %.4f value %.0f" % (sys.argv[2], j["ms_per_step"], j["value"]))
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
