#!/bin/bash
# Whole-library A/B on the GPU box: for every argument "<kernel file>[:<flags>]" (file relative to csrc/, e.g. "k_swinattn192.hip:-DW2X_A192_XRES_LATE")
# recompile that one object with the flags, relink libw2x.so and run bench.py; prints frame time and the per-kernel milliseconds of a
# frame.  An argument "base" runs the library as built.  Each variant starts from the stock objects (the previous variant's object is
# rebuilt without flags first), the last step restores the stock library.
#   tools/ab/lib_variants.sh base "k_swinattn192.hip:-DW2X_A192_XRES_LATE" base "k_pixgemm.hip@tools/ab/k_pixgemm_r2.hip"
# (an earlier revision of a kernel as the alternative source: git show <rev>:waifu2x-tensorrt_amd/csrc/k_pixgemm.hip > tools/ab/k_pixgemm_r2.hip
#  before the call - round 2's files are revision 38f61ee; the round-3 records under profiles/r3_kernels/ were taken that way)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/waifu2x-tensorrt_amd"
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
extra() { case "$1" in k_mlp2.hip) echo "-mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans";; k_mlp96q.hip|k_mlp96p.hip) echo "-mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans";; *) echo "";; esac; }
# build <object's file> <flags> [<alternative source, relative to the repository root>]
# (a host file of the library - "engine.cpp:-DW2X_EXP_..." - is compiled the way the Makefile does it: -x hip, object build/<stem>.o)
build() { local obj="build/${1%.*}.o" lang=""; case "$1" in *.cpp) lang="-x hip";; esac
          $CXX $(extra "$1") $2 -I csrc $lang -c "${3:-csrc/$1}" -o "$obj" 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; }
run() { (cd "$ROOT"; python bench.py --no-cpu-baseline --steps ${STEPS:-20} ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', '| ms/frame', d['ms_per_step'], d.get('ms_per_step_samples'), '| full path', d['config']['full_path_ms_per_frame'], '|', d['roofline']['kernels_ms_per_frame'], '| compose', d['config']['families_ms_per_frame']['compose'])"); }
last=""        # files of the previous variant, to be rebuilt stock
for arg in "$@"; do
  if [ -n "$last" ]; then for f in $last; do case "$f" in @*) rm -f "build/${f#@}";; *) build "$f" "";; esac; done; last=""; fi
  if [ "$arg" = "base" ]; then run "base"; continue; fi
  # a variant may change several files: "engine.cpp:-DX+k_mlp2.hip:-DY+k_extra.hip@tools/ab/k_extra.hip" (the last form ADDS an object that is not part of the stock library)
  ok=1; IFS='+' read -ra parts <<< "$arg"
  for part in "${parts[@]}"; do
    file=${part%%:*}; flags=""; case "$part" in *:*) flags=${part#*:};; esac
    alt=""; case "$file" in *@*) alt="$ROOT/${file#*@}"; file=${file%%@*};; esac       # "k_x.hip@tools/ab/k_x_r2.hip": that object from another source
    build "$file" "$flags" "$alt" || { echo "build failed: $part"; ok=0; }
    if [ -f "csrc/$file" ]; then last="$last $file"; else last="$last @${file%.*}.o"; fi
  done
  [ $ok = 1 ] && run "$arg"
done
if [ -n "$last" ]; then for f in $last; do case "$f" in @*) rm -f "build/${f#@}";; *) build "$f" "";; esac; done; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; fi
