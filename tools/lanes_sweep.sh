#!/bin/bash
# usage (GPU box): tools/lanes_sweep.sh OUTFILE   lanes x batch size, full-resolution bench, in-tree library
OUT=$1
run() {
  env "$@" python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$*', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})" | tee -a $OUT
}
for rep in 1 2; do
run PTC_LANES=1
run PTC_LANES=2
run PTC_LANES=1 PTC_BATCH_PATHS=536870912
run PTC_LANES=2 PTC_BATCH_PATHS=536870912
done
