#!/bin/bash
# usage: tools/envsweep.sh "<bench args>|VAR=val ..." ...
for item in "$@"; do
  ar="${item%%|*}"; ev="${item#*|}"
  env $ev python bench.py $ar --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$ar | $ev', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})"
done
