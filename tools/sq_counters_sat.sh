#!/bin/bash
# (CLX_TUNE_MERGE / CLX_TUNE_STREAMS are read by builds made with CLX_EXTRA_FLAGS="-DCLX_TUNING" only: point CLAXON_HIP_LIB at one)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out/sq_sat; mkdir -p $O
timeout 120 rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' > $O/sq_names.txt
wc -w $O/sq_names.txt
run() { name=$1; shift
  CLX_TUNE_MERGE=9 CLX_TUNE_STREAMS=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $REPO/tools/merge_probe.py 9 3 > $O/$name.log 2>&1
  python - $O/$name <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True)
if not f: print("no counters", sys.argv[1]); sys.exit()
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    if n in ("clx_k_lean","clx_k_scan"): d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in d:
    print(n, {k: "%.3g" % (sorted(v)[len(v)//2]) for k,v in d[n].items()})
PY
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH
run p3 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_WAVES SQ_LEVEL_WAVES
