#!/bin/bash
# config 4 (swin_unet/photo x4, tile 400, batch 8, TTA, 1080p) under the round-4 kernel switches:  tools/ab/cfg4_regression.sh   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
run() { env $1 python tools/op_times.py swin_unet/photo 4 3 8 400 1080 1920 tta 2>/dev/null | head -2 | tr '\n' ' ' | cut -c1-260; echo " <- $1"; }
run X=1
run W2X_A192_TWO_PER_CU=1
cd waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans -I csrc"
$CXX -DW2X_MLP96_PRIO=0 -DW2X_MLP_PREFETCH=0 -c csrc/k_mlp96q.hip -o build/k_mlp96q.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
cd ..
run MLP96_ROUND3=1
run "W2X_A192_TWO_PER_CU=1 MLP96_ROUND3=1"
