#!/usr/bin/env bash
# A/B of the empty-space-skipping frame ON the GPU box: bash tools/r02_ab_skip.sh <out> <lib names (in _var/, or "default")...>
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
for name in "$@"; do
  lib=$([ "$name" = default ] && realpath open-volume-renderer_amd/libovr_hip.so || realpath _var/libovr_hip_$name.so)
  for cs in "c3 oblique" "c3 front" "c2 oblique" "c1 oblique" "c4 oblique" "c5 oblique"; do
    set -- $cs
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --config $1 --camera $2 --skip-empty --steps 10 --warmup 3 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{}); f=d['per_frame']
print('$name $1 $2 skip', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'fetched %.1fM skipped %.1fM' % (f['samples']/1e6, f['skipped_samples']/1e6))" >> $out
  done
done
cat $out
