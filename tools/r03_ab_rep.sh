#!/usr/bin/env bash
# alternating repetitions of variant libraries on one box (the headline moves by 2 % from run to run): bash tools/r03_ab_rep.sh <out> <reps> "<bench args>" lib...
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; reps=$2; args=$3; shift 3
for rep in $(seq $reps); do
  for name in "$@"; do
    lib=$([ "$name" = default ] && realpath open-volume-renderer_amd/libovr_hip.so || realpath _var/libovr_hip_$name.so)
    OVR_HIP_LIBRARY=$lib timeout -k 10 600 python bench.py $args --no-cpu-baseline --no-views --no-skip-leg 2>/dev/null | grep "^{" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$name', '$args', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)))" >> $out
  done
done
sort $out
