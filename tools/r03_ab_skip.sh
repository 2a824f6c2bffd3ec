#!/usr/bin/env bash
# empty-space-skipping frames, A/B of variant libraries ON the GPU box: bash tools/r03_ab_skip.sh <out> <lib names in _var/ or "default">...
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
cases=${OVR_AB_CASES:-"c3:oblique:sparse c3:front:sparse c3:oblique:bumps c3:oblique:dense c2:oblique:sparse c1:oblique:sparse c5:oblique:sparse"}
scenes=${OVR_AB_SCENES:-"scene_bonsai scene_skull scene_engine scene_teapot scene_heatrelease_1 scene_vorts1.json"}
for name in "$@"; do
  lib=$([ "$name" = default ] && realpath open-volume-renderer_amd/libovr_hip.so || realpath _var/libovr_hip_$name.so)
  for cs in $cases; do
    IFS=: read cfg cam tf <<< "$cs"
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --config $cfg --camera $cam --tf $tf --skip-empty --steps 10 --warmup 5 --no-cpu-baseline --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{}); f=d['per_frame']
print('$name $cfg $cam $tf skip', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'fetched %.1f M skipped %.1f M shadow %.1f / %.1f M' % (f['samples']/1e6, f['skipped_samples']/1e6, f['shadow_samples']/1e6, f['skipped_shadow_samples']/1e6))" >> $out
  done
  for sc in $scenes; do
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python tools/scene_bench.py $sc 2>/dev/null | grep json | sed "s/^/$name /" >> $out
  done
done
cat $out
