#!/usr/bin/env bash
# A/B of variant libraries on dense cases ON the GPU box: bash tools/r02_ab_scenes.sh <out> <lib names in _var/ or "default">...
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
for name in "$@"; do
  lib=$([ "$name" = default ] && realpath open-volume-renderer_amd/libovr_hip.so || realpath _var/libovr_hip_$name.so)
  for cs in "oblique dense" "front dense" "inside sparse" "oblique sparse"; do
    set -- $cs
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --camera $1 --tf $2 --steps 10 --warmup 3 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$name c3 $1 $2', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), d['roofline']['kernel'][:34])" >> $out
  done
  for sc in scene_mechhand.json scene_vorts_t83 scene_supernova scene_lung; do
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python tools/scene_bench.py $sc 2>/dev/null | grep json | sed "s/^/$name /" >> $out
  done
done
cat $out
