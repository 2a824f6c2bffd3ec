#!/usr/bin/env bash
# round-4 profile collection ON the GPU box (run last: the traffic file is keyed by the kernel sources' hash):
#   bash tools/r04_prof_all.sh [part...]    parts: a (C3 matrix), b (C3 variants), c (C1 C2), d (C4 C5), e (rank 0's shard of 2 / 4 / 8 ranks at C3), json
set -uo pipefail
cd $GRAFT_REPO_ROOT
S="--steps 10 --warmup 3"
parts=${*:-"a b c d json"}
for part in $parts; do
  case $part in
  a)
    bash tools/prof.sh r04_c3 --config c3 $S
    bash tools/prof.sh r04_c3_front --config c3 --camera front $S
    bash tools/prof.sh r04_c3_dense --config c3 --tf dense $S
    bash tools/prof.sh r04_c3_front_dense --config c3 --camera front --tf dense $S ;;
  b)
    bash tools/prof.sh r04_c3_rate4 --config c3 --rate 4 --steps 12 --warmup 3
    bash tools/prof.sh r04_c3_fovy45 --config c3 --fovy 45 $S
    bash tools/prof.sh r04_c3_sparse --config c3 --sparse-sampling $S
    bash tools/prof.sh r04_c3_gradient --config c3 --shading 1 $S ;;
  c)
    bash tools/prof.sh r04_c1 --config c1 $S
    bash tools/prof.sh r04_c2 --config c2 $S ;;
  d)
    bash tools/prof.sh r04_c4 --config c4 --steps 6 --warmup 2
    bash tools/prof.sh r04_c5 --config c5 --steps 6 --warmup 2 ;;
  e)
    # what ONE rank of the driver's N-GPU run launches: rank 0's image shard, rendered by one process without the gather
    for w in 2 4 8; do bash tools/prof.sh r04_c3_shard_of$w --config c3 --shard-of $w $S; done ;;
  json)
    python3 tools/traffic_json.py gpurun_out/r04_traffic.json \
      "c3|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r04_c3" "c3|front|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r04_c3_front" \
      "c3|oblique|dense|2|1|1.0|60.0|0=gpurun_out/prof_r04_c3_dense" "c3|front|dense|2|1|1.0|60.0|0=gpurun_out/prof_r04_c3_front_dense" \
      "c3|oblique|sparse|2|1|4.0|60.0|0=gpurun_out/prof_r04_c3_rate4" "c3|oblique|sparse|2|1|1.0|45.0|0=gpurun_out/prof_r04_c3_fovy45" \
      "c3|oblique|sparse|2|1|1.0|60.0|1=gpurun_out/prof_r04_c3_sparse" "c3|oblique|sparse|1|1|1.0|60.0|0=gpurun_out/prof_r04_c3_gradient" \
      "c1|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r04_c1" "c2|oblique|sparse|0|1|1.0|60.0|0=gpurun_out/prof_r04_c2" \
      "c4|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r04_c4" "c5|oblique|sparse|2|1|1.0|60.0|0=gpurun_out/prof_r04_c5" \
      "c3|oblique|sparse|2|2|1.0|60.0|0=gpurun_out/prof_r04_c3_shard_of2" "c3|oblique|sparse|2|4|1.0|60.0|0=gpurun_out/prof_r04_c3_shard_of4" \
      "c3|oblique|sparse|2|8|1.0|60.0|0=gpurun_out/prof_r04_c3_shard_of8" ;;
  esac
done
