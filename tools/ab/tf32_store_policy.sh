#!/bin/bash
# Precision::TF32 frame with the fused fp32 kernels' output rows stored with the streaming cache policy (stock) against plain stores (k_f32.o rebuilt with -DW2X_ST_AUX=0), alternating.
# GPU box:  bash tools/ab/tf32_store_policy.sh [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/waifu2x-tensorrt_amd"
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; }
run() { (cd "$ROOT"; python tools/op_times.py swin_unet/art 4 3 4 256 1080 1920 tf32 2>/dev/null | grep "ms per resident frame" | sed "s/^/$1: /"); }
for r in $(seq 1 "${1:-2}"); do
  run "streaming stores"
  $CXX -DW2X_ST_AUX=0 -I csrc -c csrc/k_f32.hip -o build/k_f32.o && link && run "plain stores"
  $CXX -I csrc -c csrc/k_f32.hip -o build/k_f32.o && link
done
