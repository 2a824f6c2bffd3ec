#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of one bench configuration, kernel_stats.csv only: bash tools/stats_only.sh <tag> <bench args...>   (ON the GPU box)
set -uo pipefail
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline --no-views --no-skip-leg --no-extras > $out/stats.log 2>&1
for f in $(find $out/stats -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
rm -rf $out/stats
grep -E "raymarch|shade|composite" $out/kernel_stats.csv | cut -c1-200
