#!/bin/bash
# A/B of the shade order (creation order vs light beams, OVR_HIP_SHADE_ORDER) on the configurations whose shade kernel matters; same library, env switch.
# usage (GPU box): bash tools/r05_ab_shade_order.sh [beam sizes...]
set -uo pipefail
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r05_order; mkdir -p $o
S="--no-extras --no-cpu-baseline --no-views --no-skip-leg"
VARIANTS="${@:-32:8}"   # beam:slabs ...
CFGS=${CFGS:-all}
run() { # tag, env..., -- cfg
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py $S $@ --detail-file $o/$tag.json > /dev/null 2> $o/$tag.err || echo "$tag FAILED"
  python3 - $o/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; p = r["phase_ms_rank0"]; f = d["per_frame"]
    print(f"{sys.argv[2]:44s} ms/frame {d['ms_per_step']:.3f} march {p['march']:.3f} shade {p['shade']:.3f} comp {p['composite']:.3f} shaded {f['shaded_samples']/1e6:.2f}M shadow {f['shadow_samples']/1e6:.1f}M", flush=True)
except Exception as e:
    print(sys.argv[2], "no record:", e, flush=True)
PY
}
i=0
for cfg in "--config c3 --steps 20 --warmup 5" "--config c3 --skip-empty --steps 20 --warmup 5" "--config c3 --camera front --steps 20 --warmup 5" "--config c3 --tf dense --steps 20 --warmup 5" \
           "--config c3 --sparse-sampling --steps 20 --warmup 5" "--config c3 --fovy 45 --steps 10 --warmup 3" "--config c4 --steps 10 --warmup 3" "--config c5 --steps 8 --warmup 3" "--config c1 --steps 20 --warmup 5" \
           "--config c3 --rate 4 --steps 5 --warmup 3"; do
  i=$((i+1)); t=$(echo $cfg | tr -d ' -' | cut -c1-22)
  if [ "$CFGS" != all ] && ! echo " $CFGS " | grep -q " $i "; then continue; fi
  run ${t}_creation OVR_HIP_SHADE_ORDER=0 -- $cfg
  for v in $VARIANTS; do b=${v%%:*}; d=${v##*:}; run ${t}_beam${b}_slabs$d OVR_HIP_SHADE_ORDER=1 OVR_HIP_SHADE_BEAM=$b OVR_HIP_SHADE_SLABS=$d -- $cfg; done
done
