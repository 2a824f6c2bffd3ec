#!/usr/bin/env bash
# HERE, after `gpurun -- 'bash tools/r02_prof_all.sh; bash tools/r02_bench_all.sh'`: copy the summaries from gpurun_out/ (scratch) into profiles/ (tracked)
set -euo pipefail
cd "$(dirname "$0")/.."
for d in gpurun_out/prof_r02_*; do
  t=profiles/${d#gpurun_out/prof_}
  mkdir -p $t
  for f in kernel_stats.csv pmc_summary.txt bench.json; do [ -f $d/$f ] && cp $d/$f $t/; done
done
cp gpurun_out/r02_traffic.json profiles/r02_traffic.json
mkdir -p profiles/r02_bench
cp gpurun_out/r02_bench/*.json profiles/r02_bench/
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
t = json.load(open('profiles/r02_traffic.json'))
print('traffic hash', t['kernels_hash'], 'sources hash', bench.kernels_hash(), 'OK' if t['kernels_hash'] == bench.kernels_hash() else 'MISMATCH')
PY
