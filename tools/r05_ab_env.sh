#!/bin/bash
# env-switch A/B on one library: bash tools/r05_ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   (one line per variant; "-" = no env)
set -uo pipefail
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r05_env; mkdir -p $o
S="--no-extras --no-cpu-baseline --no-views --no-skip-leg"
cfg=$1; shift
i=0
for v in "$@"; do
  i=$((i+1)); tag=$(echo "$cfg" | tr -d ' -' | cut -c1-20)_$i
  if [ "$v" = "-" ]; then envs=(); else read -ra envs <<< "$v"; fi
  env "${envs[@]}" python bench.py $S $cfg --detail-file $o/$tag.json > /dev/null 2> $o/$tag.err || echo "$tag FAILED"
  python3 - $o/$tag.json "$cfg | $v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; p = r["phase_ms_rank0"]
    print(f"{sys.argv[2]:70s} ms/frame {d['ms_per_step']:.3f} march {p['march']:.3f} shade {p['shade']:.3f} comp {p['composite']:.3f}", flush=True)
except Exception as e:
    print(sys.argv[2], "no record:", e, flush=True)
PY
done
