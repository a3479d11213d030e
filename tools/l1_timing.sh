for t in 1 0 1 0; do
PTC_TIMING=$t PTC_LANES=1 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('timing=$t lanes=1 => %.1f Mpaths/s' % d['value'], d['ms_per_step'])"
done
PTC_TIMING=0 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('timing=0 default lanes => %.1f Mpaths/s' % d['value'], d['ms_per_step'])"
PTC_TIMING=1 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('timing=1 default lanes => %.1f Mpaths/s' % d['value'], d['ms_per_step'])"
nproc; cat /proc/cpuinfo | grep "model name" | head -1
