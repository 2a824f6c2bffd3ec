#!/usr/bin/env bash
# the round-2 tree (git worktree at _var/r02wt, built there) against this one, same box, alternating: bash tools/r03_vs_r02.sh <out> "<bench args>" ...
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1.txt; shift
for args in "$@"; do
  for rep in 1 2; do
    for tree in _var/r02wt .; do
      (cd $tree && timeout -k 10 600 python bench.py $args --no-cpu-baseline --no-views --no-skip-leg 2>/dev/null | grep "^{" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$tree', '$args', 'ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'Gs/s %.1f' % (d['value']/1e3))") >> $out
    done
  done
done
cat $out
