#!/bin/bash
# ISA of the lean kernels only (clx_lean.hip: clx_k_scan, clx_k_compose, clx_k_lean, clx_k_lean24) in ~40 s instead of the library's 1 m 45 s:
# the other files' kernels are turned into unused static functions in a scratch copy.  usage: tools/lean_isa.sh out.s [hipcc flags...]
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
T=$(mktemp -d)
mkdir -p $T/claxon_amd/csrc $T/include
cp -r $REPO/claxon_amd/csrc/* $T/claxon_amd/csrc/
cp $REPO/include/*.h $T/include/
for f in clx_kernels.hip clx_lanes.hip; do
  sed -i 's/extern "C" __global__/template <int CLX_NOT_BUILT> __global__/' $T/claxon_amd/csrc/$f
done
printf '#include "clx_kernels.hip"\n#include "clx_lanes.hip"\n#include "clx_lean.hip"\n' > $T/claxon_amd/csrc/lean_tu.hip
cd $T/claxon_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I intrin --cuda-device-only -S lean_tu.hip -o $OUT "$@" 2>&1 | grep -v "hip-link" || true
rm -rf $T
python3 - $OUT <<'PY'
import re,sys
s=open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(clx_k_\w+)\n(.*?)\.wavefront_size", s, re.S):
    d=dict(re.findall(r"\.(\w+):\s+(\d+)", m.group(2)))
    print("%-14s vgpr %3s agpr %3s sgpr %3s sgpr_spill %3s vgpr_spill %3s lds %6s scratch %s" % (m.group(1), d.get("vgpr_count"), d.get("agpr_count"), d.get("sgpr_count"), d.get("sgpr_spill_count"), d.get("vgpr_spill_count"), d.get("group_segment_fixed_size"), d.get("private_segment_fixed_size")))
PY
