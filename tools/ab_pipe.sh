#!/bin/bash
for i in 1 2 3; do for L in $1 $2; do echo -n "$L: "; CLAXON_HIP_LIB=$PWD/$L python tools/pipe_probe.py 10000 80 | grep "crc False submit"; done; done
