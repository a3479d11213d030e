#!/bin/bash
# usage (GPU box): tools/bench_specs.sh OUTFILE WORKLOAD "NAME SEGS_PER_CU" ...   short bench of prebuilt build/var/NAME/libptc.so at a given PTC_SEGMENTS_PER_CU
OUT=$1; W=$2; shift 2
for spec in "$@"; do
  set -- $spec
  PTC_SEGMENTS_PER_CU=$2 PTC_LIB=$PWD/build/var/$1/libptc.so python3 bench.py --workload $W --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$W $spec', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})" | tee -a $OUT
done
