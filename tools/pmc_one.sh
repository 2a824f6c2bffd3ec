#!/usr/bin/env bash
# one PMC pass of one bench configuration: bash tools/pmc_one.sh <tag> "<counters>" <bench args...>   (ON the GPU box; OVR_HIP_LIBRARY honoured)
set -uo pipefail
tag=$1; shift; set_=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $set_ --output-format csv -d $out/raw -- python3 bench.py "$@" --no-cpu-baseline --no-views --no-skip-leg --no-extras > $out/log.txt 2>&1 || echo "pmc failed"
python3 - <<PY
import csv, glob, collections
out="$out"
for f in sorted(glob.glob(out+"/raw/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for (k,c),(n,v) in sorted(agg.items()):
        if any(t in k for t in ("raymarch", "shade_pool", "composite")): print(f"{k} {c} n={n} mean={v/n:.6g}")
PY
rm -rf $out/raw
