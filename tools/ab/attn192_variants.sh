#!/bin/bash
# Builds tools/ab/attn192_variants from variants of csrc/k_swinattn192u.hip (or of another source via SRC=):  tools/ab/attn192_variants.sh "<flags v0>" "<flags v1>" ... ["STAMPS <flags>"]
# A last argument that starts with STAMPS builds that variant with the per-phase s_memtime stamps and prints the phase table.
# A baseline from an earlier revision: git show 38f61ee:waifu2x-tensorrt_amd/csrc/<kernel>.hip > tools/ab/<kernel>_r2.hip, then "SRC=$PWD/tools/ab/<kernel>_r2.hip".
# The retired C = 192 kernels (round 3's two-per-CU k_swinattn192_r3.hip, the 12-wave k_swinattn192w.hip) are in git history: git show 9576837:tools/ab/<file> > /tmp/<file>.
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -I $ROOT/waifu2x-tensorrt_amd/csrc -Wno-unused-function -Wno-unused-variable"
TMP=$(mktemp -d)
i=0; objs=""; hflags=""
for arg in "$@"; do
  src=$ROOT/waifu2x-tensorrt_amd/csrc/k_swinattn192u.hip; fl="$arg"
  case "$arg" in SRC=*) src=${arg%% *}; src=${src#SRC=}; fl=${arg#SRC=$src}; ;; esac
  case "$fl" in STAMPS*) fl="${fl#STAMPS} -DW2X_A192_STAMPS"; hflags="-DW2X_A192_STAMPS";; esac
  $CXX $fl -Dlaunch_swin_attn192u=launch_swin_attn96_v$i -Dlaunch_swin_attn192=launch_swin_attn96_v$i -Dlaunch_swin_attn192w=launch_swin_attn96_v$i -c "$src" -o $TMP/v$i.o      # (the wide-workgroup file names its launcher ...192w)
  objs="$objs $TMP/v$i.o"; i=$((i+1))
done
$CXX -DNVAR=$i -DCW=192 $hflags -c $ROOT/tools/ab/attn96_variants.hip -o $TMP/main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $TMP/main.o $objs -o $ROOT/tools/ab/attn192_variants
rm -rf $TMP
echo "built tools/ab/attn192_variants with $i variants"
