#!/usr/bin/env bash
# the foveated (sparse-sampling) C3 frame with the interactive app's default focus, A/B over libraries / switches ON the GPU box
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --sparse-sampling --steps 20 --warmup 5 --no-cpu-baseline --no-skip-leg --no-views 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline'].get('phase_ms_rank0',{}); f=d['per_frame']
print('ms %.3f' % d['ms_per_step'], 'march %.3f shade %.3f comp %.3f' % (p.get('march',0),p.get('shade',0),p.get('composite',0)), 'px %d Msamples %.1f shaded %.1f shadow %.1f chunks %d' % (f['active_pixels'], f['samples']/1e6, f['shaded_samples']/1e6, f['shadow_samples']/1e6, d['roofline']['pool_chunks']), d['config']['volume_layout_read'])"; }
for a in "$@"; do run $a; done
