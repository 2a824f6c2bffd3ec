#!/usr/bin/env bash
# Builds the UNMODIFIED reference host pipeline (apps/main_batch.cpp + scene/serializer/imageio/dylink)
# from the sources where they lie under $OVR_ROOT (default /root/reference) into oracle/_ref/.
# Nothing from the reference is copied into this repository; oracle/_ref/ is git-ignored.
#
# Products:
#   oracle/_ref/renderbatch        the reference's headless app (apps/main_batch.cpp:240-318), no built-in devices;
#                                  it resolves `--device hip` through dlopen("libdevice_hip.so") (ovr/renderer.cpp:55-58)
#   oracle/_ref/plugin_probe       oracle/plugin_probe.cpp: drives libdevice_hip.so through the reference's MainRenderer interface
#                                  (sparse sampling, spp, accumulation, camera change, swap)
#   oracle/_ref/libovr_refhost.so  scene.cpp + serializer + imageio + colormaps, used by tests to pin the
#                                  oracle's scene/TF/PNG-quantisation restatement against the real reference code
#
# The OptiX and OSPRay devices are NOT buildable here (no nvcc/optix.h/libospray) - see DESIGN.md.
set -euo pipefail
R="${OVR_ROOT:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
OBJ="${OVR_REF_OBJDIR:-/tmp/ovr_ref_obj}"
if [ ! -d "$R/ovr" ]; then echo "[build_ref] $R not present - skipping (prebuilt oracle/_ref is used if it exists)"; exit 0; fi
mkdir -p "$OUT" "$OBJ"
INC="-I$R -I$R/ovr -I$R/ovr/common -I$R/extern -I$R/extern/gdt -I$R/extern/tfn/colormaps -I$R/apps"
CXX="${CXX:-g++}"
FLAGS="-std=c++17 -O2 -fPIC -w $INC"
SRCS="ovr/scene.cpp ovr/renderer.cpp ovr/serializer/serializer_vidi3d.cpp ovr/serializer/serializer_diva.cpp \
ovr/common/imageio.cpp ovr/common/generate_mask.cpp ovr/common/dylink/Library.cpp"
CM=$(cd "$R" && ls extern/tfn/colormaps/colormap.cpp extern/tfn/colormaps/*/*.cpp)
objs=""
pids=""
n=0
for s in $SRCS $CM apps/main_batch.cpp; do
  o="$OBJ/$(echo "$s" | tr '/' '_').o"
  objs="$objs $o"
  if [ ! -f "$o" ] || [ "$R/$s" -nt "$o" ]; then
    $CXX $FLAGS -c "$R/$s" -o "$o" &
    n=$((n+1)); if [ $((n % 6)) -eq 0 ]; then wait; fi
  fi
done
wait
hostobjs=$(echo "$objs" | tr ' ' '\n' | grep -v main_batch | tr '\n' ' ')
$CXX -shared -o "$OUT/libovr_refhost.so" $hostobjs -ldl -lpthread
$CXX -o "$OUT/renderbatch" $objs -rdynamic -ldl -lpthread
$CXX $FLAGS "$HERE/ref_probe.cpp" -o "$OUT/ref_probe" -L"$OUT" -lovr_refhost -Wl,-rpath,'$ORIGIN' -ldl -lpthread
$CXX $FLAGS "$HERE/ref_probe_wide.cpp" -o "$OUT/ref_probe_wide" -L"$OUT" -lovr_refhost -Wl,-rpath,'$ORIGIN' -ldl -lpthread
$CXX $FLAGS "$HERE/ref_scene_probe.cpp" -o "$OUT/ref_scene_probe" -L"$OUT" -lovr_refhost -Wl,-rpath,'$ORIGIN' -ldl -lpthread
$CXX $FLAGS "$HERE/plugin_probe.cpp" -o "$OUT/plugin_probe" -L"$OUT" -lovr_refhost -Wl,-rpath,'$ORIGIN' -rdynamic -ldl -lpthread
echo "[build_ref] built $OUT/renderbatch, $OUT/libovr_refhost.so, $OUT/ref_probe, $OUT/ref_scene_probe and $OUT/plugin_probe"
