#!/bin/bash
# single-lane breakdown of the bench (kernel times do not overlap, so they add up to the wall time)
export PTC_LANES=1
python bench.py --steps 8 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('1 lane => %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})"
