#!/usr/bin/env bash
# what the measured choice of layout and pipeline settles on (OVR_HIP_TUNE_TRACE) ON the GPU box: bash tools/r03_tune.sh <out>
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt
cases=${OVR_AB_CASES:-"oblique:sparse:4 oblique:dense:1 oblique:sparse:1 front:dense:4 front:dense:1 oblique:dense:4"}
scenes=${OVR_AB_SCENES:-"scene_lung scene_supernova scene_mechhand.json scene_vorts_t83 scene_body scene_zebrafish scene_chameleon scene_bonsai"}
rates=${OVR_AB_RATES:-"1.0 4.0"}
for cs in $cases; do
  IFS=: read cam tf rate <<< "$cs"
  echo "== c3 $cam $tf rate $rate" >> $out
  OVR_HIP_TUNE_TRACE=1 timeout -k 10 300 python bench.py --camera $cam --tf $tf --rate $rate --steps 5 --warmup 14 --no-cpu-baseline --no-skip-leg --no-views 2> $out.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('tuned: ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'read', d['config']['volume_layout_read'], d['roofline']['kernel'][:60])" >> $out
  grep "tune:" $out.err >> $out
  OVR_HIP_TUNE=0 timeout -k 10 300 python bench.py --camera $cam --tf $tf --rate $rate --steps 5 --warmup 3 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('rules: ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'read', d['config']['volume_layout_read'], d['roofline']['kernel'][:60])" >> $out
done
for sc in $scenes; do
  for rate in $rates; do
    echo "== $sc rate $rate" >> $out
    OVR_HIP_TUNE_TRACE=1 OVR_SCENE_RATE=$rate OVR_SCENE_SKIP_LEG=0 OVR_SCENE_WARMUP=14 timeout -k 10 600 python tools/scene_bench.py $sc 2> $out.err | grep json | sed "s/^/tuned /" >> $out
    grep "tune: decided" $out.err >> $out
    OVR_HIP_TUNE=0 OVR_SCENE_RATE=$rate OVR_SCENE_SKIP_LEG=0 timeout -k 10 600 python tools/scene_bench.py $sc 2>/dev/null | grep json | sed "s/^/rules /" >> $out
  done
done
cat $out
