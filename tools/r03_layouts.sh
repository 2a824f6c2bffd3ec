#!/usr/bin/env bash
# one library, the layout a frame reads forced (0 general, 1 thin, 2 thin transposed, 3 quad, -1 automatic) ON the GPU box:
#   bash tools/r03_layouts.sh <out> "<layouts>" ; OVR_AB_CASES ("cam:tf:rate[:extra bench args]") / OVR_AB_SCENES override the lists
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
layouts=$1; shift
cases=${OVR_AB_CASES:-"oblique:sparse:4 oblique:dense:1 oblique:sparse:1 front:dense:4 front:sparse:1"}
scenes=${OVR_AB_SCENES:-"scene_lung scene_supernova scene_mechhand.json scene_vorts_t83 scene_body scene_zebrafish"}
for l in $layouts; do
  for cs in $cases; do
    IFS=: read cam tf rate extra <<< "$cs"
    timeout -k 10 300 python bench.py --camera $cam --tf $tf --rate $rate --layout $l ${extra:-} --steps 5 --warmup 2 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{}); f=d['per_frame']
print('layout $l c3 $cam $tf rate $rate ${extra:-}', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'read', d['config']['volume_layout_read'], d['roofline']['kernel'][:34])" >> $out
  done
  for sc in $scenes; do
    OVR_SCENE_SKIP_LEG=0 OVR_SCENE_LAYOUT=$l timeout -k 10 300 python tools/scene_bench.py $sc 2>/dev/null | grep json | sed "s/^/layout $l /" >> $out
  done
done
cat $out
