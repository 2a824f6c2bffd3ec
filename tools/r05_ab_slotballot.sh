set -uo pipefail
cd $GRAFT_REPO_ROOT
o=gpurun_out/r05_slot; mkdir -p $o
S="--no-extras --no-cpu-baseline --no-views --no-skip-leg --skip-empty"
for rep in 1 2; do
for lib in open-volume-renderer_amd/libovr_hip.so _var/libovr_hip_slotballot.so; do
  for cfg in "--config c3 --steps 20 --warmup 5" "--config c3 --rate 4 --steps 6 --warmup 3" "--config c3 --camera front --steps 20 --warmup 5" "--config c4 --steps 10 --warmup 3" "--config c1 --steps 20 --warmup 5"; do
    tag=$(basename $lib .so)_$(echo $cfg | tr -d ' -' | cut -c1-24)_$rep
    OVR_HIP_LIBRARY=$PWD/$lib python bench.py $S $cfg --detail-file $o/$tag.json > /dev/null 2> $o/$tag.err || echo "$tag FAILED"
    python3 - $o/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; p = r["phase_ms_rank0"]; f = d["per_frame"]
    print(f"{sys.argv[2]:52s} ms/frame {d['ms_per_step']:.3f} march {p['march']:.3f} shade {p['shade']:.3f} comp {p['composite']:.3f} shadow fetched {f['shadow_samples']/1e6:.1f}M skipped {f['skipped_shadow_samples']/1e6:.1f}M", flush=True)
except Exception as e:
    print(sys.argv[2], "no record:", e, flush=True)
PY
  done
done
done
OVR_HIP_LIBRARY=$PWD/_var/libovr_hip_slotballot.so python -m pytest tests/test_parity_gpu.py tests/test_round2_gpu.py tests/test_shipped_scenes_gpu.py tests/test_config_sweep_gpu.py -q -k "not bench and not stand_in and not two_ranks" 2>&1 | tail -3
