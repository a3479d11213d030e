#!/bin/bash
# usage (GPU box): tools/floor_variants.sh OUTFILE NAME...   launch_floor.py for prebuilt build/var/NAME/libptc.so, one lane
OUT=$1; shift
for name in "$@"; do echo "== $name" | tee -a $OUT; PTC_LANES=1 PTC_LIB=$PWD/build/var/$name/libptc.so python3 tools/launch_floor.py 2>&1 | tee -a $OUT; done
