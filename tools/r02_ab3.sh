#!/usr/bin/env bash
# A/B of variant libraries over "config camera layout" triples ON the GPU box: OVR_AB_CASES="c4 oblique 0,c4 front -1" bash tools/r02_ab3.sh <out> <libs...>
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
IFS=',' read -ra cases <<< "${OVR_AB_CASES:-c4 oblique 0}"
for name in "$@"; do
  lib=$(realpath _var/libovr_hip_$name.so)
  for cs in "${cases[@]}"; do
    set -- $cs
    OVR_HIP_LIBRARY=$lib timeout -k 10 400 python bench.py --config $1 --camera $2 --layout $3 --steps 6 --warmup 2 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$name $1 $2 layout $3 ->', d['config']['volume_layout_read'], 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'pipe', round(d['roofline']['pipeline']['frac'],3))" >> $out
  done
done
cat $out
