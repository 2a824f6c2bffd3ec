#!/usr/bin/env bash
# round-5 profile collection ON the GPU box (run last: the traffic file is keyed by the kernel sources' hash):
#   bash tools/r05_prof_all.sh [part...]    parts: default (the default command under --stats), a (C3 matrix), b (C3 variants), c (C1 C2), d (C4 C5), e (rank 0's shard of 2 / 4 / 8 ranks at C3), json
set -uo pipefail
cd $GRAFT_REPO_ROOT
S="--steps 10 --warmup 3"
parts=${*:-"a b c d json"}
for part in $parts; do
  case $part in
  a)
    bash tools/prof.sh r05_c3 --config c3 $S
    bash tools/prof.sh r05_c3_front --config c3 --camera front $S
    bash tools/prof.sh r05_c3_dense --config c3 --tf dense $S
    bash tools/prof.sh r05_c3_front_dense --config c3 --camera front --tf dense $S ;;
  b)
    bash tools/prof.sh r05_c3_rate4 --config c3 --rate 4 --steps 12 --warmup 3
    bash tools/prof.sh r05_c3_fovy45 --config c3 --fovy 45 $S
    bash tools/prof.sh r05_c3_sparse --config c3 --sparse-sampling $S
    bash tools/prof.sh r05_c3_gradient --config c3 --shading 1 $S ;;
  c)
    bash tools/prof.sh r05_c1 --config c1 $S
    bash tools/prof.sh r05_c2 --config c2 $S ;;
  d)
    bash tools/prof.sh r05_c4 --config c4 --steps 6 --warmup 2
    bash tools/prof.sh r05_c5 --config c5 --steps 6 --warmup 2 ;;
  default)
    # the DEFAULT command as the driver runs it (every leg), under rocprofv3 --kernel-trace --stats: the stdout line, the detail record and the per-kernel statistics
    mkdir -p gpurun_out/prof_r05_default_cmd
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_default_cmd/stats -- python3 bench.py --steps 20 --warmup 5 --detail-file gpurun_out/prof_r05_default_cmd/bench_detail.json > gpurun_out/prof_r05_default_cmd/bench_stdout.json 2> gpurun_out/prof_r05_default_cmd/stats.log
    for f in $(find gpurun_out/prof_r05_default_cmd/stats -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/prof_r05_default_cmd/kernel_stats.csv; done
    rm -rf gpurun_out/prof_r05_default_cmd/stats
    wc -c gpurun_out/prof_r05_default_cmd/bench_stdout.json; head -5 gpurun_out/prof_r05_default_cmd/kernel_stats.csv | cut -c1-200 ;;
  e)
    # what ONE rank of the driver's N-GPU run launches: rank 0's image shard, rendered by one process without the gather
    for w in 2 4 8; do bash tools/prof.sh r05_c3_shard_of$w --config c3 --shard-of $w $S; done ;;
  json)
    # (the parts run in several gpurun calls, each on a fresh box: run `json` where every gpurun_out/prof_r05_* has been merged back - no GPU needed)
    python3 tools/traffic_json.py gpurun_out/r05_traffic.json \
      "c3|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r05_c3" "c3|front|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r05_c3_front" \
      "c3|oblique|dense|2|1|1.0|60.0|0=gpurun_out/prof_r05_c3_dense" "c3|front|dense|2|1|1.0|60.0|0=gpurun_out/prof_r05_c3_front_dense" \
      "c3|oblique|sparse|2|1|4.0|60.0|0=gpurun_out/prof_r05_c3_rate4" "c3|oblique|sparse|2|1|1.0|45.0|0=gpurun_out/prof_r05_c3_fovy45" \
      "c3|oblique|sparse|2|1|1.0|60.0|1=gpurun_out/prof_r05_c3_sparse" "c3|oblique|sparse|1|1|1.0|60.0|0=gpurun_out/prof_r05_c3_gradient" \
      "c1|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r05_c1" "c2|oblique|sparse|0|1|1.0|60.0|0=gpurun_out/prof_r05_c2" \
      "c4|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r05_c4" "c5|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r05_c5" \
      "c3|oblique|sparse|2|2|1.0|60.0|0=gpurun_out/prof_r05_c3_shard_of2" "c3|oblique|sparse|2|4|1.0|60.0|0=gpurun_out/prof_r05_c3_shard_of4" \
      "c3|oblique|sparse|2|8|1.0|60.0|0=gpurun_out/prof_r05_c3_shard_of8" ;;
  esac
done
