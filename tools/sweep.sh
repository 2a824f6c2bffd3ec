#!/usr/bin/env bash
# camera/config sweep ON the GPU box: bash tools/sweep.sh <out-file> [configs...]
out=$1; shift
cfgs=${@:-"c2 c3g c3"}
for c in $cfgs; do for cam in front oblique; do
  timeout -k 10 200 python bench.py --config $c --camera $cam --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_frame']
print('$c $cam', round(d['value']), 'Msamp/s', round(d['ms_per_step'],2), 'ms  kern', round(d['roofline']['pipeline']['kernel_ms'],2), 'samples %.1fM shaded %.2fM shadow %.1fM' % (p['samples']/1e6, p['shaded_samples']/1e6, p['shadow_samples']/1e6), 'phases', {k: round(v,2) for k,v in d['roofline']['phase_ms_rank0'].items()}, 'frac', round(d['roofline']['pipeline']['frac'],4), 'nominal', round(d['roofline']['pipeline']['nominal_frac_survey_F4'],4))" >> $out
done; done
cat $out
