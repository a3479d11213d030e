#!/bin/bash
# usage (GPU box): tools/viewer_variants.sh OUTFILE [VAR=val ...] -- NAME...   the viewer's frame loop (tools/viewer_loop.py, 1080p and 720p at 1 spp, 1080p at 4 spp) on prebuilt build/var/NAME/libptc.so
OUT=$1; shift
ENVS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS+=("$1"); shift; done
shift
for name in "$@"; do
  for cfg in "1 40 1920 1080" "1 40 1280 720" "4 30 1920 1080"; do
    env "${ENVS[@]}" PTC_LIB=$PWD/build/var/$name/libptc.so python3 tools/viewer_loop.py atrium $cfg 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$name ${ENVS[*]}', d['w'], d['h'], 'x%d' % d['spp'], 'ms/frame %.2f' % d['ms_per_frame']['median'], 'refit %.2f' % d['ms_update_and_refit'], 'Mpaths/s %.0f' % d['Mpaths_per_s'])" | tee -a $OUT
  done
done
