#!/usr/bin/env bash
# bench lines of every BASELINE configuration on one GPU (ON the GPU box); copied to profiles/ afterwards
set -uo pipefail
cd $GRAFT_REPO_ROOT
o=gpurun_out/r02_bench; mkdir -p $o
timeout -k 10 600 python bench.py > $o/bench_c3_default.json 2> $o/bench_c3_default.err
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $o/bench_c2.json 2>/dev/null
timeout -k 10 300 python bench.py --config c1 --no-cpu-baseline > $o/bench_c1.json 2>/dev/null
timeout -k 10 600 python bench.py --config c4 --no-cpu-baseline --no-views > $o/bench_c4_1gpu.json 2>/dev/null
timeout -k 10 600 python bench.py --config c5 --steps 64 --warmup 2 --no-cpu-baseline --no-views > $o/bench_c5_1gpu.json 2>/dev/null
timeout -k 10 600 python bench.py --config c5tea --steps 16 --warmup 2 --no-cpu-baseline --no-views > $o/bench_c5tea_1gpu.json 2>/dev/null
timeout -k 10 300 python bench.py --camera front --no-cpu-baseline --no-views > $o/bench_c3_front.json 2>/dev/null
OVR_BENCH_FORCE_GATHER=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --no-cpu-baseline > $o/bench_c3_forced_gather_rccl.json 2> $o/forced_gather.err
OVR_BENCH_BACKEND=gloo OVR_BENCH_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 10 > $o/bench_c3_2ranks_one_card_gloo.json 2> $o/two_ranks.err
for f in $o/*.json; do python3 - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1])
    print("$f".split("/")[-1], "value %.0f Msamples/s  fps %.1f  ms %.3f  frac %.3f  pipe %.3f  n_gpus %d" % (d["value"], d["fps"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["pipeline"]["frac"], d["n_gpus"]))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
