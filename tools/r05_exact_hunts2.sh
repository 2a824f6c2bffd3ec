#!/bin/bash
# Wide one-off hunts with EQUALITY as the bar on the FINAL kernels of round 5 (parity instrument: libovr_hip_parity.so + the oracle's det pow), with the shade order by light beams (default)
# and the 16- / 8-bit layouts' row loads FORCED (OVR_HIP_ROW_LOADS=1: the size rule never takes them on test-sized volumes).  ON the GPU box.
set -uo pipefail
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r05_hunts2; mkdir -p $o
export OVR_HIP_LIBRARY=$PWD/open-volume-renderer_amd/libovr_hip_parity.so OVR_PARITY_EXACT_RUN=1 OVR_ORACLE_POWF=det
for rows in 1 0; do
  export OVR_HIP_ROW_LOADS=$rows
  for seed in 521 522; do
    echo "== row loads $rows, sweep seed $seed"
    OVR_DETPOW_SEED=$seed OVR_DETPOW_CASES=500 timeout -k 10 900 python tests/parity_exact_check.py sweep 2>&1 | tail -2
  done
  echo "== row loads $rows, old generator's sweep under new seeds"
  OVR_SWEEP_SEED=77$rows OVR_SWEEP_CASES=300 timeout -k 10 900 python -m pytest tests/test_config_sweep_gpu.py -q 2>&1 | tail -2
  echo "== row loads $rows, state-machine fuzzer (plain, non-finite voxels + lazy replicas, group of 3)"
  timeout -k 10 600 python tests/fuzz_states.py 80 5$rows 12 2>&1 | tail -1
  OVR_FUZZ_NONFINITE=1 OVR_FUZZ_LAZY=1 timeout -k 10 600 python tests/fuzz_states.py 60 6$rows 12 2>&1 | tail -1
  OVR_FUZZ_GROUP=3 OVR_HIP_QUIET=1 timeout -k 10 600 python tests/fuzz_states.py 40 7$rows 10 2>&1 | tail -1
done 2>&1 | tee $o/exact_hunts2.txt
