#!/usr/bin/env bash
# on the GPU box: timing, then FETCH_SIZE / request counters per kernel of tools/gather_granule.cpp
set -uo pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/granule; mkdir -p $out; rm -f $out/timing.txt
B=/tmp/gather_granule; hipcc --offload-arch=gfx950 -O2 tools/gather_granule.cpp -o $B || exit 1
for g in 16 0.0625 0.002; do timeout -k 10 120 $B $g >> $out/timing.txt 2>&1 || exit 1; done
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$n -- $B 16 > $out/pmc_$n.log 2>&1 || echo "failed: $set" >> $out/errors.log
done
python3 - <<PY
import csv, glob, collections
out="$out"
with open(out+"/pmc_summary.txt","w") as o:
    for f in sorted(glob.glob(out+"/**/*counter_collection.csv", recursive=True)):
        agg=collections.defaultdict(lambda: [0,0.0])
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])][0]+=1; agg[(r["Kernel_Name"][:40], r["Counter_Name"])][1]+=float(r["Counter_Value"])
        for (k,c),(n,v) in sorted(agg.items()):
            o.write(f"{k} {c} dispatches={n} mean={v/n:.6g}\n")
PY
rm -rf $out/pmc_*/
cat $out/timing.txt $out/pmc_summary.txt
