#!/bin/bash
# A/B of library builds in the saturated regime: ONE merged launch of M runs at a time (tools/merge_probe.py) under a kernel trace;
# prints each kernel's median duration.  (CLX_TUNE_MERGE / CLX_TUNE_STREAMS are read by builds made with -DCLX_TUNING only.)  usage: tools/gpu_ab_sat.sh "A B" rounds M   (builds as for tools/gpu_ab.sh; resolves ~1 %)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
VARS=${1:-"A B"}; ROUNDS=${2:-2}; M=${3:-9}
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out/absat; mkdir -p $O
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    export CLAXON_HIP_LIB=$REPO/claxon_amd/libclaxon_hip_$v.so
    rm -rf $O/$v.$r
    CLX_TUNE_MERGE=$M CLX_TUNE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$v.$r -o t -- python $REPO/tools/merge_probe.py $M 4 > $O/$v.$r.log 2>&1
    python - $O/$v.$r "$v r$r M$M" <<'PY'
import csv,glob,sys,statistics as st
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
d={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n.startswith("clx_k_"): d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("%-10s " % sys.argv[2] + "  ".join("%s %.0f" % (k[6:], st.median(v[-4:])) for k,v in d.items() if st.median(v)>20), " exact:", open(sys.argv[1]+".log").read().count("exact: True"))
PY
  done
done
