#!/usr/bin/env bash
# profiling helper run ON the GPU box: bash tools/prof.sh <tag> <bench args...>
# (the bench runs with --no-views --no-skip-leg: only the named configuration launches kernels, so per-kernel means are its own)
set -uo pipefail
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline --no-views --no-skip-leg --no-extras > $out/stats.log 2>&1
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$n -- python3 bench.py "$@" --no-cpu-baseline --no-views --no-skip-leg --no-extras > $out/pmc_$n.log 2>&1 || echo "pmc set failed: $set" >> $out/errors.log
done
python3 - <<PY
import csv, glob, os, collections
out="$out"
for f in sorted(glob.glob(out+"/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    with open(out+"/pmc_summary.txt","a") as o:
        for (k,c),(n,v) in sorted(agg.items()):
            if any(t in k for t in ("raymarch", "shade_pool", "composite", "relayout", "macrocell", "minmax")): o.write(f"{k} {c} dispatches={n} mean={v/n:.6g}\n")
for f in sorted(glob.glob(out+"/stats/**/*kernel_stats.csv", recursive=True)):
    os.system(f"cp {f} {out}/kernel_stats.csv")
PY
# keep the summaries, drop the raw traces (gpurun merges at most 64 MiB back)
rm -rf $out/stats $out/pmc_*/
