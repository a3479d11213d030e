#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh TAG [bench args]  -> gpurun_out/pmcq_TAG/pmc_summary.json: three counter passes (instructions / lanes / waits, HBM fetch, HBM write)
# of `bench.py --steps 3 --warmup 1`, each in its own run (never combined with a trace).  The quick look between experiments; tools/profile.sh is the full set.
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o pmc -- $B > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
done
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > $OUT/pmc_summary.json
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -delete
python3 - <<PY
import json
d=json.load(open('$OUT/pmc_summary.json'))
bench=None
for line in open('$OUT/pmc1.log'):
    if line.startswith('{') and '"metric"' in line: bench=json.loads(line)
scale=(bench['steps']+bench['warmup'])/bench['steps']
for k,v in d.items():
    for kn in ('k_trace_closest','k_trace_any','k_shade'):
        if kn in k and 'raster' not in k and v.get('SQ_INSTS_VALU'):
            u=bench['kernels'][kn]; n=u['units_per_launch']*u['launches']*scale
            print(kn, 'valu/unit %.2f'%(v['SQ_INSTS_VALU']/n), 'lane_util %.2f'%v.get('_valu_lane_util',0), 'wait_mem %.2f'%v.get('_wait_any_frac',0), 'wait_issue %.2f'%v.get('_wait_inst_frac',0),
                  'fetch B/unit %.1f'%(v.get('FETCH_SIZE',0)*2048/n), 'write B/unit %.1f'%(v.get('WRITE_SIZE',0)*1024/n), 'ms/launch %.2f'%u['avg_launch_ms'])
PY
