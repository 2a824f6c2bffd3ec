#!/usr/bin/env bash
# A/B of variant libraries ON the GPU box: bash tools/r02_ab.sh <out-name> <lib names in _var/ ...>
# per library: a parity smoke (tests/test_parity_gpu.py) and C3 front / oblique + C2 bench lines
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
for name in "$@"; do
  lib=$(realpath _var/libovr_hip_$name.so)
  if [ "${OVR_AB_PARITY:-1}" = "1" ]; then
    OVR_HIP_LIBRARY=$lib timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q > gpurun_out/ab_parity_$name.log 2>&1
    echo "$name parity: $(tail -1 gpurun_out/ab_parity_$name.log)" >> $out
  fi
  IFS=',' read -ra cfgs <<< "${OVR_AB_CONFIGS:-c3 front,c3 oblique,c2 oblique}"
  for cfg in "${cfgs[@]}"; do
    set -- $cfg
    OVR_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --config $1 --camera $2 --steps 10 --warmup 3 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$name $1 $2', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'frac', round(d['roofline']['frac'],3), 'pipe', round(d['roofline']['pipeline']['frac'],3))" >> $out
  done
done
cat $out
