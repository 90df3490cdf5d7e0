#!/bin/bash
# (CLX_TUNE_MERGE / CLX_TUNE_STREAMS are read by builds made with CLX_EXTRA_FLAGS="-DCLX_TUNING" only: point CLAXON_HIP_LIB at one)
# lean / scan durations of ONE merged launch of M runs at a time (no other launch beside it), M = 1..12
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out/launch_width; mkdir -p $O
for M in 1 2 3 4 6 8 9 10 12; do
  CLX_TUNE_MERGE=$M CLX_TUNE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$M -o t -- python $REPO/tools/merge_probe.py $M 4 > $O/m$M.log 2>&1
  python - $O/m$M $M <<'PY'
import csv,glob,sys,statistics as st
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
d={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n.startswith("clx_k_"): d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
M=int(sys.argv[2])
print("M %2d  " % M + "  ".join("%s %.0f us (%.0f/run)" % (k[6:], st.median(v[-4:]), st.median(v[-4:])/M) for k,v in d.items() if st.median(v)>20))
PY
  tail -2 $O/m$M.log | head -1
done
