#!/usr/bin/env bash
# A/B of an environment switch of libovr_hip.so ON the GPU box: bash tools/r02_ab_env.sh <out> <VAR> "<values>"   (cases: OVR_AB_CASES="c3 oblique sparse,...")
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; var=$2; vals=$3
IFS=',' read -ra cases <<< "${OVR_AB_CASES:-c3 oblique sparse}"
for v in $vals; do
  for cs in "${cases[@]}"; do
    set -- $cs
    env $var=$v timeout -k 10 300 python bench.py --config $1 --camera $2 --tf $3 --steps 10 --warmup 3 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$var=$v $1 $2 $3', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)))" >> $out
  done
done
cat $out
