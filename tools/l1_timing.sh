#!/bin/bash
# Diagnostic (GPU box): single-lane wall time vs. the sum of its kernel times, with and without timing events and at two batch sizes.
# On some boxes of the pool a one-lane run shows wall >> kernel sum (the GPU idles between launches); two lanes hide it.
run() {
  env "$@" python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$*', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})"
}
run PTC_LANES=1 PTC_TIMING=1
run PTC_LANES=1 PTC_TIMING=0
run PTC_LANES=1 PTC_BATCH_PATHS=134217728
run PTC_LANES=1 PTC_BATCH_PATHS=33554432
run PTC_LANES=2
run PTC_LANES=3
