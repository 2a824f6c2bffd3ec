#!/usr/bin/env bash
# round-2 profile collection ON the GPU box (run last: the traffic file is keyed by the kernel sources' hash)
set -uo pipefail
cd $GRAFT_REPO_ROOT
S="--steps 10 --warmup 3"
bash tools/prof.sh r02_c3 --config c3 $S
bash tools/prof.sh r02_c3_front --config c3 --camera front $S
bash tools/prof.sh r02_c3_dense --config c3 --tf dense $S
bash tools/prof.sh r02_c3_front_general --config c3 --camera front --layout 0 $S
bash tools/prof.sh r02_c2 --config c2 $S
bash tools/prof.sh r02_c2_front --config c2 --camera front --layout 0 $S
bash tools/prof.sh r02_c2_front_lds --config c2 --camera front --lds-staging --layout 0 $S
# the default command itself (views matrix, skipping leg, CPU baseline off): kernel stats only
mkdir -p gpurun_out/prof_r02_default_cmd
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_default_cmd/stats -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_r02_default_cmd/bench.json 2> gpurun_out/prof_r02_default_cmd/stats.log
cp $(find gpurun_out/prof_r02_default_cmd/stats -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r02_default_cmd/kernel_stats.csv
python3 tools/traffic_json.py gpurun_out/r02_traffic.json "c3|oblique|sparse|2|1=gpurun_out/prof_r02_c3" "c3|front|sparse|2|1=gpurun_out/prof_r02_c3_front" \
  "c3|oblique|dense|2|1=gpurun_out/prof_r02_c3_dense" "c2|oblique|sparse|0|1=gpurun_out/prof_r02_c2"
rm -rf gpurun_out/prof_r02_default_cmd/stats
