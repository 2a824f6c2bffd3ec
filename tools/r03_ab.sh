#!/usr/bin/env bash
# round-3 A/B of variant libraries ON the GPU box, the all-shaded / rate-4 regime first:
#   bash tools/r03_ab.sh <out> <lib names in _var/ or "default">...      (OVR_AB_CASES / OVR_AB_SCENES override the lists)
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
cases=${OVR_AB_CASES:-"oblique:sparse:4 oblique:dense:1 oblique:sparse:1 front:dense:4"}
scenes=${OVR_AB_SCENES:-"scene_lung scene_supernova scene_mechhand.json"}
for name in "$@"; do
  lib=$([ "$name" = default ] && realpath open-volume-renderer_amd/libovr_hip.so || realpath _var/libovr_hip_$name.so)
  for cs in $cases; do
    IFS=: read cam tf rate <<< "$cs"
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --camera $cam --tf $tf --rate $rate --steps 5 --warmup 2 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{}); f=d['per_frame']
print('$name c3 $cam $tf rate $rate', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'Msamples %.1f shaded %.1f shadow %.1f' % (f['samples']/1e6, f['shaded_samples']/1e6, f['shadow_samples']/1e6), d['roofline']['kernel'][:34])" >> $out
  done
  for sc in $scenes; do
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python tools/scene_bench.py $sc 2>/dev/null | grep json | sed "s/^/$name /" >> $out
  done
done
cat $out
