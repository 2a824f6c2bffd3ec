#!/usr/bin/env bash
# bench lines of every BASELINE configuration + the rehearsals of the N > 1 paths ON the GPU box (one card): bash tools/r04_bench_all.sh
set -uo pipefail
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_bench; mkdir -p $o
python bench.py > $o/bench_c3_default.json 2> $o/bench_c3_default.err
python bench.py --config c2 > $o/bench_c2.json 2>/dev/null
python bench.py --config c1 > $o/bench_c1.json 2>/dev/null
python bench.py --rate 4 --steps 5 --warmup 2 --no-cpu-baseline --no-views --no-skip-leg > $o/bench_c3_rate4.json 2>/dev/null
python bench.py --tf dense --camera front --rate 4 --steps 5 --warmup 2 --no-cpu-baseline --no-views --no-skip-leg > $o/bench_c3_front_dense_rate4.json 2>/dev/null
OVR_BENCH_FORCE_GATHER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $o/bench_c3_forced_gather_rccl.json 2> $o/fg.err
OVR_BENCH_BACKEND=gloo OVR_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $o/bench_c3_2ranks_one_card_gloo.json 2> $o/2r.err
python bench.py --devices 0,0,0,0 --steps 10 --warmup 5 --no-cpu-baseline --no-views --no-skip-leg > $o/bench_c3_device_group_4_on_one_card.json 2>/dev/null
for f in $o/*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], "ms %.3f fps %.1f Gs/s %.1f" % (d["ms_per_step"], d["fps"], d["value"]/1e3), "bound", r["bound"], "frac", None if r["frac"] is None else round(r["frac"],3), "traffic", r["traffic"] and round(r["traffic"]/1e9,2), "util", r.get("utilisation"), "tuning", d["config"].get("tuning"), d["config"]["volume_layout_read"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
python tools/renderbatch_c3.py > $o/renderbatch_reference_app.txt 2>&1; cat $o/renderbatch_reference_app.txt | grep renderbatch_c3
