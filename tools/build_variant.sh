#!/usr/bin/env bash
# builds a variant of libovr_hip.so with extra -D flags into _var/ (git-ignored, travels to the GPU box): tools/build_variant.sh <name> [flags...]
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _var/obj_$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $*"
/opt/rocm/bin/hipcc $F -c open-volume-renderer_amd/csrc/ovr_hip_kernels.hip -o _var/obj_$name/k.o &
/opt/rocm/bin/hipcc $F -x hip -c open-volume-renderer_amd/csrc/ovr_hip_api.cpp -o _var/obj_$name/a.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _var/libovr_hip_$name.so _var/obj_$name/k.o _var/obj_$name/a.o
echo built _var/libovr_hip_$name.so
