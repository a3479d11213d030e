#!/bin/bash
# usage (on the GPU box): tools/profile.sh TAG [bench args]   → gpurun_out/prof_TAG/{lane1.txt, kernel_stats.csv, pmc_summary.json, bench*.json}
# One-lane breakdown, kernel trace of the default configuration, then counter passes (each in its own run, kernel-trace/stats never combined with --pmc).
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
PTC_LANES=1 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null > $OUT/bench_lane1.json
python3 -c "import json; d=json.load(open('$OUT/bench_lane1.json')); print('1 lane => %.1f Mpaths/s' % d['value'], {k: round(v,4) for k,v in d['seconds'].items()})" | tee $OUT/lane1.txt
# the kernel trace runs bench.py's DEFAULT command (16 steps, 2 warm-up): its per-kernel average durations are the ones the bench line's
# `avg_launch_ms` must agree with; the counter passes below use 3 steps (they serialise the dispatches and take longer)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 || exit 1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do   # (the TA_* counters abort rocprofv3 on this image)
  i=$((i+1))
  echo "pmc pass $i: $set"
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o pmc -- $B > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 > $OUT/pmc_summary.json
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -delete
python3 - <<PY
import json
d=json.load(open('$OUT/pmc_summary.json'))
for k,v in d.items():
    if 'trace' in k or 'shade' in k:
        g=v.get('GRBM_GUI_ACTIVE',0)
        print(k[:40], 'disp',v.get('dispatches'), 'VALU %.3g'%v.get('SQ_INSTS_VALU',0), 'SALU %.3g'%v.get('SQ_INSTS_SALU',0), 'VMEM_RD %.3g'%v.get('SQ_INSTS_VMEM_RD',0), 'LDS %.3g'%v.get('SQ_INSTS_LDS',0),
              'GUI %.4g'%g, 'TA_BUSY %.3g'%v.get('TA_TA_BUSY_sum',0), 'wait_any %.2f'%v.get('_wait_any_frac',0), 'wait_inst %.2f'%v.get('_wait_inst_frac',0), 'lane_util %.2f'%v.get('_valu_lane_util',0))
PY
