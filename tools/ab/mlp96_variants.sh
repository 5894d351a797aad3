#!/bin/bash
# Builds tools/ab/mlp96_variants (here or on the GPU box) from variants of csrc/k_mlp96q.hip:  tools/ab/mlp96_variants.sh "<flags v0>" "<flags v1>" ...
# e.g.  tools/ab/mlp96_variants.sh "" "-DW2X_MLP_EXP=1" "-DW2X_MLP_EXP=2"      then run tools/ab/mlp96_variants on the GPU box
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -I $ROOT/waifu2x-tensorrt_amd/csrc -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans"
SRC=${MLP96_SRC:-$ROOT/waifu2x-tensorrt_amd/csrc/k_mlp96q.hip}
TMP=$(mktemp -d)
i=0; objs=""
for fl in "$@"; do
  $CXX $fl -Dlaunch_mlp96q=launch_mlp96_v$i -Dmlp96q_supported=mlp96_supported_v$i -c "$SRC" -o $TMP/v$i.o
  objs="$objs $TMP/v$i.o"; i=$((i+1))
done
$CXX -DNVAR=$i -c $ROOT/tools/ab/mlp96_variants.hip -o $TMP/main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $TMP/main.o $objs -o $ROOT/tools/ab/mlp96_variants
rm -rf $TMP
echo "built tools/ab/mlp96_variants with $i variants"
