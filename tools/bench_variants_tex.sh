#!/bin/bash
# like bench_variants.sh, for the textured workload (BASELINE configs[4])
OUT=$1; shift
ENVS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS+=("$1"); shift; done
shift
for name in "$@"; do
  env "${ENVS[@]}" PTC_LIB=$PWD/build/var/$name/libptc.so python3 bench.py --workload textured --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('textured $name ${ENVS[*]}', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})" | tee -a $OUT
done
