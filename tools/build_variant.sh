#!/usr/bin/env bash
# builds a variant of libovr_hip.so with extra -D flags into _var/ (git-ignored, travels to the GPU box): tools/build_variant.sh <name> [flags...]
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _var/obj_$name
make -s -C open-volume-renderer_amd/csrc -j${OVR_VARIANT_JOBS:-7} OBJDIR=$PWD/_var/obj_$name OUT=$PWD/_var/libovr_hip_$name.so EXTRA="$*"
echo built _var/libovr_hip_$name.so
