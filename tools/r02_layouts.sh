#!/usr/bin/env bash
# layout x camera matrix of one config ON the GPU box: bash tools/r02_layouts.sh <out-name> <config> "<cams>" "<layouts>"
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; cfg=$2
for cam in $3; do for ly in $4; do
  timeout -k 10 400 python bench.py --config $cfg --camera $cam --layout $ly --steps 8 --warmup 2 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$cfg $cam layout $ly ->', d['config']['volume_layout_read'], 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'frac', round(d['roofline']['frac'],3), 'pipe', round(d['roofline']['pipeline']['frac'],3), 'Msamp/s', round(d['value']))" >> $out
done; done
cat $out
