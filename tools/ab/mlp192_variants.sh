#!/bin/bash
# Builds tools/ab/mlp192_variants from variants of csrc/k_mlp2.hip (C = 192 geometry switches):  tools/ab/mlp192_variants.sh "<flags v0>" "<flags v1>" ...
# e.g.  tools/ab/mlp192_variants.sh "" "-DW2X_MLP192_TT=1 -DW2X_MLP192_WPS=3"        then run tools/ab/mlp192_variants [rows] on the GPU box
# FRAG32_MASK=0b10 (environment): bit i set = variant i gets its weights in the 32x32x16 fragment order (the mlp2q kernel)
# A baseline from an earlier revision: git show 38f61ee:waifu2x-tensorrt_amd/csrc/<kernel>.hip > tools/ab/<kernel>_r2.hip, then "SRC=$PWD/tools/ab/<kernel>_r2.hip".
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -I $ROOT/waifu2x-tensorrt_amd/csrc -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans"      # (the Makefile's flags for k_mlp2.o; -fhonor-nans as a variant flag gives round 5's build)
SRC=${MLP192_SRC:-$ROOT/waifu2x-tensorrt_amd/csrc/k_mlp2.hip}
TMP=$(mktemp -d)
i=0; objs=""
for arg in "$@"; do
  src=$SRC; fl="$arg"
  case "$arg" in SRC=*) src=${arg%% *}; src=${src#SRC=}; fl=${arg#SRC=$src}; ;; esac      # "SRC=<file> <flags>": another source for this variant
  $CXX $fl -Dlaunch_mlp2=launch_mlp96_v$i -c "$src" -o $TMP/v$i.o
  objs="$objs $TMP/v$i.o"; i=$((i+1))
done
$CXX -fno-honor-nans -c $ROOT/waifu2x-tensorrt_amd/csrc/k_mlp96q.hip -o $TMP/q.o
$CXX -DNVAR=$i -DCW=192 -DFRAG32_MASK=${FRAG32_MASK:-0} -c $ROOT/tools/ab/mlp96_variants.hip -o $TMP/main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $TMP/main.o $TMP/q.o $objs -o $ROOT/tools/ab/mlp192_variants
rm -rf $TMP
echo "built tools/ab/mlp192_variants with $i variants"
