#!/bin/bash
# Round 6's measurements on the GPU box (stages chained in one gpurun call).  Output under gpurun_out/r06/.
#   tests                        the GPU suite (pytest -m gpu)
#   ab   rounds "cfgs" [lib]     pipelined steps (bench.py), the pool's tickets against round 5's two kernels (--pool on / off) alternating on
#                                this box; cfgs of: c3d (the driver's 20 steps) c3 (48) c3l (96) c2 c5 c4
#   line [args]                  the bench line as the driver takes it (default arguments + args) -> r06/bench_line*.json
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
O=$REPO/gpurun_out/r06; mkdir -p $O
stage=$1; shift
case $stage in
tests)
  cd $REPO; timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15 | tee $O/tests.txt ;;
ab)
  # variants: build[:pool] -- build "head" = claxon_amd/libclaxon_hip.so, X = libclaxon_hip_X.so; pool on (default) or off
  ROUNDS=${1:-2}; CFGS=${2:-"c3d c3l"}; VARS=${3:-"head:off head:on"}
  cd $REPO
  for r in $(seq 1 $ROUNDS); do for v in $VARS; do
    lib=${v%%:*}; pool=on; [ "$v" != "${v#*:}" ] && pool=${v#*:}
    if [ "$lib" = head ]; then unset CLAXON_HIP_LIB; else export CLAXON_HIP_LIB=$REPO/claxon_amd/libclaxon_hip_$lib.so; fi
    for name in $CFGS; do
      case $name in
        c3)  args="--steps 48 --warmup 6" ;;
        c3d) args="--steps 20 --warmup 5" ;;
        c3l) args="--steps 96 --warmup 8" ;;
        c2)  args="--workload config2 --steps 48" ;;
        c5)  args="--workload config5 --total-frames 80000 --unique 4096 --shard-of 8 --shard-rank 3 --steps 48" ;;
        c4)  args="--workload config4 --steps 48" ;;
      esac
      f=$O/ab_${lib}_$pool.$name.$r
      timeout 300 python bench.py --no-cpu-baseline --no-extras --pool $pool $args > $f.json 2> $f.err
      python - "$f.json" "$lib pool=$pool $name r$r" <<'PY' | tee -a $O/ab.txt
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m=j["roofline"].get("merged_launch") or {}
    print("%-28s ms/step %.4f (min %.4f max %.4f)  merged %s" % (sys.argv[2], j["ms_per_step"], j.get("ms_per_step_min", 0), j.get("ms_per_step_max", 0), {k: round(v,3) for k,v in (m.get("kernel_ms") or {}).items() if v > 0.02}))
except Exception as e: print("ERR", sys.argv[2], e, open(sys.argv[1][:-5]+".err").read()[-600:])
PY
    done
  done; done ;;
trace)
  # kernel trace of the pipelined steps: trace "variants" steps
  VARS=${1:-"head"}; STEPS=${2:-48}
  cd /tmp && export TMPDIR=/tmp
  for v in $VARS; do
    lib=${v%%:*}; pool=off; [ "$v" != "${v#*:}" ] && pool=${v#*:}
    if [ "$lib" = head ]; then unset CLAXON_HIP_LIB; else export CLAXON_HIP_LIB=$REPO/claxon_amd/libclaxon_hip_$lib.so; fi
    rm -rf $O/trace_$lib.$pool
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$lib.$pool -o t -- python $REPO/bench.py --steps $STEPS --warmup 12 --no-cpu-baseline --no-extras --repeats 3 --pool $pool > $O/trace_$lib.$pool.log 2>&1
    python $REPO/tools/trace_pipelined.py $O/trace_$lib.$pool all > $O/trace_$lib.$pool.$STEPS.txt 2>&1
    rm -rf $O/trace_$lib.$pool
  done ;;
m12pmc)
  # HBM counters of ONE merged launch of twelve runs at a time (tools/merge_probe.py 12 3), separate --pmc passes: m12pmc "variants"
  VARS=${1:-"head"}
  cd /tmp && export TMPDIR=/tmp
  for v in $VARS; do
    if [ "$v" = head ]; then unset CLAXON_HIP_LIB; else export CLAXON_HIP_LIB=$REPO/claxon_amd/libclaxon_hip_$v.so; fi
    for set in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/m12_${v}_$set
      timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/m12_${v}_$set -o p -- python $REPO/tools/merge_probe.py 12 3 > $O/m12_${v}_$set.log 2>&1
      python - $O/m12_${v}_$set "$v $set" <<'PY' | tee -a $O/m12pmc.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True)
d={}
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    if n.startswith("clx_k_"): d.setdefault(n,[]).append(float(r["Counter_Value"]))
print("%-16s" % sys.argv[2], {k[6:]: round(sum(v)/len(v)/12/1024, 1) for k,v in d.items() if sum(v)/len(v) > 1000}, "(MiB-units per run: FETCH x 2 = MB fetched)")
PY
      rm -rf $O/m12_${v}_$set
    done
  done ;;
line)
  cd $REPO
  timeout 600 python bench.py "$@" > $O/bench_line.json 2> $O/bench_line.err; tail -c 1500 $O/bench_line.json ;;
esac
