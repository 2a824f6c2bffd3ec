#!/usr/bin/env bash
# Round 5: brick shapes of the general u16 layout at C4 (2048^3 u16, rays 3.4 - 4.8 voxels apart: the frame reads the whole volume, so the apron's share of the bytes counts).
# Variants built by tools/build_variant.sh into _var/.  ON the GPU box: bash tools/r05_u16_bricks.sh
set -uo pipefail
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_u16; mkdir -p $o
S="--steps 12 --warmup 3 --no-extras --no-cpu-baseline --no-views --no-skip-leg --layout 0"
for rep in 1 2; do
for cam in oblique oblique_y diagonal front; do
  for lib in open-volume-renderer_amd/libovr_hip.so _var/libovr_hip_u16_7x4x2.so _var/libovr_hip_u16_7x2x4.so; do
    tag=$(basename $lib .so)_${cam}_$rep
    OVR_HIP_LIBRARY=$PWD/$lib python bench.py $S --config c4 --camera $cam --detail-file $o/$tag.json > /dev/null 2> $o/$tag.err || echo "$tag FAILED"
    python3 - $o/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; p = r["phase_ms_rank0"]
    print(f"{sys.argv[2]:40s} ms/frame {d['ms_per_step']:.3f} march {p['march']:.3f} shade {p['shade']:.3f} composite {p['composite']:.3f} resident {r['volume_resident_bytes'] / 1e9:.1f} GB samples {d['per_frame']['samples']:.0f}", flush=True)
except Exception as e:
    print(sys.argv[2], "no record:", e, flush=True)
PY
  done
done
done
