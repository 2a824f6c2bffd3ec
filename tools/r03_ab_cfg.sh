#!/usr/bin/env bash
# A/B of variant libraries over whole BASELINE configurations ON the GPU box: bash tools/r03_ab_cfg.sh <out> "<cfg[:extra args] ...>" <lib names in _var/ or "default">...
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
cfgs=$1; shift
for name in "$@"; do
  lib=$([ "$name" = default ] && realpath open-volume-renderer_amd/libovr_hip.so || realpath _var/libovr_hip_$name.so)
  for cs in $cfgs; do
    IFS=: read cfg extra <<< "$cs"
    OVR_HIP_LIBRARY=$lib timeout -k 10 600 python bench.py --config $cfg ${extra//,/ } --steps 10 --warmup 3 --no-cpu-baseline --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{}); f=d['per_frame']; s=d.get('with_empty_space_skipping') or {}
print('$name $cfg ${extra:-}', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'Gsamples/s %.1f' % (d['value']/1e3), 'skip-leg ms %.3f' % s.get('ms_per_step', 0), d['config']['volume_layout_read'], d['roofline']['kernel'][:30])" >> $out
  done
done
cat $out
