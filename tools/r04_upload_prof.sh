#!/usr/bin/env bash
# on the GPU box: kernel times of the upload / replica-build kernels (rocprofv3 --stats) for C3 (general, thin transposed, quad) and C4
set -uo pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/upload_prof; mkdir -p $out; rm -f $out/summary.txt
run() { # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 bench.py "$@" --steps 4 --warmup 3 --no-extras --no-cpu-baseline --no-views --no-skip-leg > $out/$tag.log 2>&1 || echo "failed: $tag" >> $out/summary.txt
  python3 - <<PY >> $out/summary.txt
import csv, glob
for f in glob.glob("$out/$tag/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ovrhip" in r["Name"] and any(t in r["Name"] for t in ("relayout", "macrocell_range", "minmax", "axis_tables")):
            print("$tag", r["Name"][:90], "calls", r["Calls"], "avg_ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
  rm -rf $out/$tag
}
run c3_rate4 --rate 4
run c3_front --camera front
run c4 --config c4
cat $out/summary.txt
