#!/bin/bash
# Round 5's measurements on the GPU box, one parameterised script (stages chained in one gpurun call; builds X are
# claxon_amd/libclaxon_hip_X.so, made beforehand with CLAXON_HIP_LIB=... CLX_EXTRA_FLAGS=... python -c "import claxon_amd as cx; cx.build(force=True)").
#   coissue                      tools/ubench/coissue (vector + scalar / LDS / branch: additive or co-issued?)
#   sat  "A B" M rounds          one merged launch of M runs at a time under a kernel trace: each kernel's median duration per build
#   pipe "A B" rounds [cfgs]     pipelined steps (bench.py, 48 steps and the driver's 20), builds alternating; cfgs of: c3 c3d c5 c4 c2
#   pad  A "0 2600 11000"        clx_k_lean with extra dynamic LDS per wave (fewer waves per CU): pipelined config-3 step per pad
#   sq   A M                     SQ busy / wait / instruction counters of clx_k_lean and clx_k_scan in a merged launch of M runs
# Output under gpurun_out/r05/.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
O=$REPO/gpurun_out/r05; mkdir -p $O
lib() { echo $REPO/claxon_amd/libclaxon_hip_$1.so; }
stage=$1; shift
case $stage in
coissue)
  timeout 90 $REPO/tools/ubench/coissue | tee $O/coissue.txt ;;
sat)
  VARS=$1; M=${2:-8}; ROUNDS=${3:-2}
  cd /tmp && export TMPDIR=/tmp
  for r in $(seq 1 $ROUNDS); do for v in $VARS; do
    export CLAXON_HIP_LIB=$(lib $v); rm -rf $O/sat_$v.$r
    CLX_TUNE_MERGE=$M CLX_TUNE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/sat_$v.$r -o t -- python $REPO/tools/merge_probe.py $M 4 > $O/sat_$v.$r.log 2>&1
    python - $O/sat_$v.$r "$v r$r M$M" <<'PY' | tee -a $O/sat.txt
import csv,glob,sys,statistics as st
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
d={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n.startswith("clx_k_"): d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("%-10s " % sys.argv[2] + "  ".join("%s %.0f" % (k[6:], st.median(v[-4:])) for k,v in d.items() if st.median(v)>10), " exact:", open(sys.argv[1]+".log").read().count("exact: True"))
PY
    rm -rf $O/sat_$v.$r
  done; done ;;
pipe)
  VARS=$1; ROUNDS=${2:-2}; CFGS=${3:-"c3 c3d"}
  cd $REPO
  for r in $(seq 1 $ROUNDS); do for v in $VARS; do
    export CLAXON_HIP_LIB=$(lib $v)
    for name in $CFGS; do
      case $name in
        c3)  args="--steps 48 --warmup 6" ;;
        c3d) args="--steps 20 --warmup 5" ;;
        c3l) args="--steps 96 --warmup 8" ;;
        c2)  args="--workload config2 --steps 48" ;;
        c5)  args="--workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48" ;;
        c4)  args="--workload config4 --steps 48" ;;
      esac
      timeout 300 python bench.py --no-cpu-baseline --no-extras $args > $O/pipe_$v.$name.$r.json 2> $O/pipe_$v.$name.$r.err
      python - "$O/pipe_$v.$name.$r.json" "$v $name r$r" <<'PY' | tee -a $O/pipe.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s ms/step %.4f (min %.4f max %.4f)  alone %s" % (sys.argv[2], j["ms_per_step"], j.get("ms_per_step_min", 0), j.get("ms_per_step_max", 0), {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items() if v > 0.05}))
except Exception as e: print("ERR", sys.argv[2], e)
PY
    done
  done; done ;;
pad)
  v=$1; PADS=$2
  cd $REPO; export CLAXON_HIP_LIB=$(lib $v)
  for r in 1 2; do for p in $PADS; do
    CLX_LEAN_LDS_PAD=$p timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 48 --warmup 6 > $O/pad_$p.$r.json 2> $O/pad_$p.$r.err
    python - "$O/pad_$p.$r.json" "pad $p r$r" <<'PY' | tee -a $O/pad.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s ms/step %.4f  alone %s" % (sys.argv[2], j["ms_per_step"], {k: round(v,3) for k,v in j["roofline"].get("kernel_ms",{}).items() if v > 0.05}))
except Exception as e: print("ERR", sys.argv[2], e)
PY
  done; done ;;
sq)
  v=$1; M=${2:-12}
  cd /tmp && export TMPDIR=/tmp; export CLAXON_HIP_LIB=$(lib $v)
  run() { name=$1; shift
    CLX_TUNE_MERGE=$M CLX_TUNE_STREAMS=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/sq_$name -o p -- python $REPO/tools/merge_probe.py $M 3 > $O/sq_$name.log 2>&1
    python - $O/sq_$name "$*" <<'PY' | tee -a $O/sq.txt
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True)
if not f: print("no counters", sys.argv[1]); sys.exit()
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    if n in ("clx_k_lean","clx_k_scan"): d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in d:
    print(n, {k: "%.4g" % (sorted(v)[len(v)//2]) for k,v in d[n].items()})
PY
    rm -rf $O/sq_$name
  }
  echo "build $v, one merged launch of $M runs at a time" | tee -a $O/sq.txt
  timeout 120 rocprofv3 -L > $O/avail.txt 2>&1; grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' > $O/sq_names.txt
  run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
  run p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
  run p3 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD ;;
esac
