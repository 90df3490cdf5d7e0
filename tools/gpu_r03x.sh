#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03x; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -o /tmp/storeshape tools/ubench/storeshape.hip 2> $O/build.err
for n in 20000 180000; do timeout 120 /tmp/storeshape $n | tee $O/storeshape_$n.txt; done
