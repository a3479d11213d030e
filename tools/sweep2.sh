#!/bin/bash
# usage: tools/sweep2.sh "<EXTRA flags>|<ENV assignments>" ...
for item in "$@"; do
  ex="${item%%|*}"; ev="${item#*|}"
  make -C physically-based-renderer_amd/csrc EXTRA="$ex" -B > /dev/null 2>&1 || { echo "build failed: $ex"; continue; }
  env $ev python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$ex | $ev', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})"
done
