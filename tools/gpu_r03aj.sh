#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/profile_r03.sh config3 config5 config4 2>&1 | grep -v "^clx_k\|^$" | tail -12
