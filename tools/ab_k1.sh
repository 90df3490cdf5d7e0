#!/bin/bash
# A/B of two builds of the library on the SAME box (box-to-box variation is ~2 %): kernel durations of the bench workload, one batch at
# a time, alternating between the builds.  usage: tools/ab_k1.sh libA.so libB.so [rounds]
R=${3:-3}
for i in $(seq $R); do
  for L in $1 $2; do
    echo -n "$L: "; CLAXON_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 40 | grep -o "kernel_ms\": {[^}]*}"
  done
done
