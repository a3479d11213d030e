#!/bin/bash
# usage (GPU box): tools/shade_variants.sh OUTFILE   config 5's k_shade (bench.py --workload textured) in four forms, each with its kernel event sum and, from three
# rocprofv3 --pmc passes, lane utilisation and HBM-side bytes per segment: no sort / the 8-class sort / two rings (textured | rest) / the environment sample's loads issued by every lane
OUT=$1
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
run() {   # name lib env...
  name=$1; lib=$2; shift 2
  env "$@" PTC_LIB=$PWD/build/var/$lib/libptc.so python3 bench.py --workload textured --steps 6 --warmup 1 --no-cpu-baseline --direct-scene 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); k=d['kernels']['k_shade']; print('$name', '=> %.1f Mpaths/s' % d['value'], 'k_shade %.2f ms per launch, %.4f s per 6 steps' % (k['avg_launch_ms'], d['seconds']['shade']))" | tee -a $OUT
  D=gpurun_out/pmcq_$name; mkdir -p $D; i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    i=$((i+1))
    env "$@" PTC_LIB=$PWD/build/var/$lib/libptc.so timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $D/pmc$i -o pmc -- python3 bench.py --workload textured --steps 3 --warmup 1 --no-cpu-baseline --direct-scene > $D/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $D/pmc$i.log; }
  done
  python3 tools/pmc_summary.py $D/pmc1 $D/pmc2 $D/pmc3 > $D/pmc_summary.json
  find $D -name '*.csv' -size +2M -delete; find $D -name '*.db' -delete
  python3 - <<PY | tee -a $OUT
import json
d=json.load(open('$D/pmc_summary.json'))
bench=None
for line in open('$D/pmc1.log'):
    if line.startswith('{') and '"metric"' in line: bench=json.loads(line)
scale=(bench['steps']+bench['warmup'])/bench['steps']
for k,v in d.items():
    if 'k_shade' in k and 'raster' not in k and v.get('SQ_INSTS_VALU'):
        u=bench['kernels']['k_shade']; n=u['units_per_launch']*u['launches']*scale
        print('   $name: lane utilisation %.2f' % v.get('_valu_lane_util',0), 'waiting memory %.2f issue %.2f' % (v.get('_wait_any_frac',0), v.get('_wait_inst_frac',0)),
              'HBM-side fetch %.0f + write %.0f B per segment' % (v.get('FETCH_SIZE',0)*2048/n, v.get('WRITE_SIZE',0)*1024/n), 'VALU %.1f per segment' % (v['SQ_INSTS_VALU']/n))
PY
}
run nosort sbase
run sort8 sbase PTC_SHADE_SORT=1
run tworing tworing PTC_SHADE_SORT=1
run envalways envalways
