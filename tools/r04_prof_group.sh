#!/usr/bin/env bash
# rocprofv3 kernel statistics of the in-process device group on ONE card (four members on device 0): bash tools/r04_prof_group.sh
set -uo pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r04_group; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/stats -- python3 bench.py --devices 0,0,0,0 --steps 10 --warmup 5 --no-cpu-baseline --no-views --no-skip-leg --no-extras > $out/stats.log 2>&1
for f in $(find $out/stats -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
for f in $(find $out/stats -name "*memory_copy_stats.csv"); do cp $f $out/memory_copy_stats.csv; done
rm -rf $out/stats
head -12 $out/kernel_stats.csv | cut -c1-160; cat $out/memory_copy_stats.csv 2>/dev/null | head -5
