#!/usr/bin/env bash
# Round 5: C4 (2048^3 u16) on one GPU - the layouts the tuner never tries on it (its march is not shade-heavy), and the supertile order of the
# launch list - each with the frame's phases and, for the candidates, FETCH_SIZE.  ON the GPU box: bash tools/r05_c4_experiments.sh
set -uo pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r05_c4; mkdir -p $o
S="--steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-views --no-skip-leg"
run() { # tag, env..., -- bench args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py $S "$@" --detail-file $o/$tag.json > /dev/null 2> $o/$tag.err || echo "$tag FAILED"
  python3 - $o/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; p = r["phase_ms_rank0"]
    print(f"{sys.argv[2]:28s} ms/frame {d['ms_per_step']:.3f} march {p['march']:.3f} shade {p['shade']:.3f} composite {p['composite']:.3f} layout {d['config']['volume_layout_read']} upload {r.get('upload_ms')} resident {r['volume_resident_bytes'] / 1e9:.1f} GB", flush=True)
except Exception as e:
    print(sys.argv[2], "no record:", e, flush=True)
PY
}
run c4_base X=1 -- --config c4
run c4_morton OVR_HIP_SCHED_ORDER=morton -- --config c4
run c4_quad X=1 -- --config c4 --layout 3
run c4_thin X=1 -- --config c4 --layout 1
run c4_thin_t X=1 -- --config c4 --layout 2
run c3_base X=1 -- --config c3
run c3_morton OVR_HIP_SCHED_ORDER=morton -- --config c3
run c2_base X=1 -- --config c2
run c2_morton OVR_HIP_SCHED_ORDER=morton -- --config c2
# counters: memory-side read requests of the march and the shade kernel, base vs Z-curve order
for tag in c4_base c4_morton; do
  ord=row; [ $tag = c4_morton ] && ord=morton
  OVR_HIP_SCHED_ORDER=$ord rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_$tag -- python3 bench.py --config c4 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-views --no-skip-leg --detail-file $o/pmc_$tag.json > $o/pmc_$tag.log 2>&1
  python3 - $o/pmc_$tag $tag <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
        if r["Counter_Name"] == "FETCH_SIZE" and k in ("raymarch_kernel", "shade_pool_kernel", "composite_kernel"):
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
print(sys.argv[2], "FETCH_SIZE x 2 KiB -> GB per launch:", {k: round(v[1] / v[0] * 2 * 1024 / 1e9, 2) for k, v in agg.items()}, flush=True)
PY
  rm -rf $o/pmc_$tag
done
