#!/bin/bash
# usage: tools/envsweep.sh "VAR=val VAR2=val" ...   runs a short bench under each environment
for ev in "$@"; do
  env $ev python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$ev', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})"
done
