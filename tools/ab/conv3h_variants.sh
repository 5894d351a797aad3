#!/bin/bash
# A/B of the few-output-channel 3x3 kernel on the GPU box: each argument "<label>=<source relative to the repo>[:<flags>]" is compiled into build/k_conv3h.o,
# the library relinked and config 2 (cunet/art x2, 1080p) timed per op; the stock object is restored at the end.
#   tools/ab/conv3h_variants.sh r4=tools/ab/k_conv3h_r4.hip new=waifu2x-tensorrt_amd/csrc/k_conv3h.hip new3=waifu2x-tensorrt_amd/csrc/k_conv3h.hip:-DW2X_C3H_WPC2=3
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/waifu2x-tensorrt_amd"
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
cp build/k_conv3h.o /tmp/k_conv3h.stock.o
for round in 1 2; do
for arg in "$@"; do
  label=${arg%%=*}; rest=${arg#*=}; src=${rest%%:*}; flags=""; case "$rest" in *:*) flags=${rest#*:};; esac
  $CXX $flags -I csrc -c "$ROOT/$src" -o build/k_conv3h.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o || { echo "build failed: $arg"; continue; }
  (cd "$ROOT"; timeout -k 10 150 python tools/op_times.py cunet/art 2 1 4 256 1080 1920 2>/dev/null | grep -E "ms per resident|N=16 |N=4 " | cut -c1-70 | sed "s/^/$label  /")
done
done
cp /tmp/k_conv3h.stock.o build/k_conv3h.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
