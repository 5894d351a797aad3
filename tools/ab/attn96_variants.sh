#!/bin/bash
# Builds tools/ab/attn96_variants (here or on the GPU box) from variants of the C = 96 attention kernel:
#   tools/ab/attn96_variants.sh "<src or -> [flags] v0" "<flags v1>" ...   each argument: optional "SRC=<file> " prefix, then compiler flags
# e.g.  tools/ab/attn96_variants.sh "" "SRC=/tmp/k_swinattn96_other.hip"      (the build switches of rounds 3 - 5 - W2X_A96_PV32 / BUF / BQ_LDS / XRES_EARLY / WPS / PRIO - are resolved
#        in the shipped file since round 6; the file that still has them: git show 9576837:waifu2x-tensorrt_amd/csrc/k_swinattn96.hip)
# A baseline from an earlier revision: git show 38f61ee:waifu2x-tensorrt_amd/csrc/<kernel>.hip > tools/ab/<kernel>_r2.hip, then "SRC=$PWD/tools/ab/<kernel>_r2.hip".
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -I $ROOT/waifu2x-tensorrt_amd/csrc -Wno-unused-function -Wno-unused-variable"
TMP=$(mktemp -d)
i=0; objs=""
for arg in "$@"; do
  src=$ROOT/waifu2x-tensorrt_amd/csrc/k_swinattn96.hip; fl="$arg"
  case "$arg" in SRC=*) src=${arg%% *}; src=${src#SRC=}; fl=${arg#SRC=$src}; ;; esac
  $CXX $fl -Dlaunch_swin_attn96=launch_swin_attn96_v$i -c "$src" -o $TMP/v$i.o
  objs="$objs $TMP/v$i.o"; i=$((i+1))
done
$CXX -DNVAR=$i -c $ROOT/tools/ab/attn96_variants.hip -o $TMP/main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $TMP/main.o $objs -o $ROOT/tools/ab/attn96_variants
rm -rf $TMP
echo "built tools/ab/attn96_variants with $i variants"
