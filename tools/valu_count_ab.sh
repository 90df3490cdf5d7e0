#!/bin/bash
# Executed instructions of clx_k_lean / clx_k_scan for two builds of the library (claxon_amd/libclaxon_hip_X.so, see tools/r05_gpu.sh):
# rocprofv3 --pmc SQ_INSTS_VALU ... of bench.py's config-3 steps one at a time.  usage (through gpurun): tools/valu_count_ab.sh [P N]
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for v in ${*:-P N}; do
  export CLAXON_HIP_LIB=$REPO/claxon_amd/libclaxon_hip_$v.so
  rm -rf /tmp/pm_$v
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm_$v -o p -- \
      python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pipeline --path lanes-fused > /tmp/pm_$v.log 2>&1
  python - /tmp/pm_$v $v <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True)
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    if n in ("clx_k_lean","clx_k_scan"): d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in d: print(sys.argv[2], n, {k: "%.5g" % (sorted(v)[len(v)//2]) for k,v in d[n].items()})
PY
done
