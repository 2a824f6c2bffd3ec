#!/bin/bash
# A/B bench of differently built libraries: tools/ab.sh <bench args> -- lib1.so lib2.so ...   (one JSON summary line per library)
args=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done; shift
for lib in "$@"; do
  OVR_HIP_LIBRARY=$(realpath $lib) timeout -k 10 300 python bench.py --no-cpu-baseline "${args[@]}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{})
print('$lib', round(d['ms_per_step'],4), 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'frac', round(d['roofline']['pipeline']['frac'],3))"
done
