#!/usr/bin/env bash
# round-2 diagnosis ON the GPU box: view x TF matrix of C3, per-wave trace of the front view, PMC profile of the front view
set -uo pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_diag; mkdir -p $out
for cam in front oblique; do for tf in sparse dense; do
  timeout -k 10 200 python bench.py --config c3 --camera $cam --tf $tf --steps 10 --warmup 3 --no-cpu-baseline --no-skip-leg 2>/dev/null | tail -1 > $out/c3_${cam}_${tf}.json
  python3 - <<PY >> $out/matrix.txt
import json
d=json.load(open("$out/c3_${cam}_${tf}.json")); p=d['per_frame']; r=d['roofline']
print("c3 $cam $tf", round(d['value']), 'Msamp/s', round(d['ms_per_step'],3), 'ms samples %.1fM shaded %.2fM shadow %.1fM' % (p['samples']/1e6, p['shaded_samples']/1e6, p['shadow_samples']/1e6), {k: round(v,3) for k,v in r['phase_ms_rank0'].items()}, 'march frac', round(r['kernels']['raymarch_kernel']['frac'],3), 'pipe frac', round(r['pipeline']['frac'],3))
PY
done; done
cat $out/matrix.txt
for cam in front oblique; do
  OVR_HIP_TRACE=1 OVR_HIP_TRACE_FILE=$out/trace_$cam.bin timeout -k 10 200 python bench.py --config c3 --camera $cam --steps 2 --warmup 1 --no-cpu-baseline --no-skip-leg > /dev/null 2>&1
  echo "== trace $cam" >> $out/trace.txt
  python3 tools/trace_analyze.py $out/trace_$cam.bin >> $out/trace.txt 2>&1
done
cat $out/trace.txt
bash tools/prof.sh r02_front --config c3 --camera front --steps 5 --warmup 2 --no-skip-leg
cat gpurun_out/prof_r02_front/pmc_summary.txt
