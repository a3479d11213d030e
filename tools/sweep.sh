#!/bin/bash
# usage: tools/sweep.sh "<EXTRA flags>" ...  (short full-res bench per build)
for ex in "$@"; do
  make -C physically-based-renderer_amd/csrc EXTRA="$ex" -B > /dev/null 2>&1 || { echo "build failed: $ex"; continue; }
  python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$ex', '=> %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})"
done
