#!/bin/bash
# A/B on one box: the stem folded into the patch convolution's launch (engine.cpp fuse_stem, the default) against the two launches (W2X_NO_FUSE_STEM=1).
# Alternating bench.py processes; prints ms per frame, the full-path figure and the conv48 / stem kernel sums of each.
# usage (on the GPU box): bash tools/ab/fuse_stem_ab.sh [rounds]  > gpurun_out/<dir>/fuse_stem_ab.txt
cd "$(dirname "$0")/../.." || exit 1
rounds=${1:-3}
for r in $(seq 1 "$rounds"); do
  for v in fused unfused; do
    if [ $v = unfused ]; then export W2X_NO_FUSE_STEM=1; else unset W2X_NO_FUSE_STEM; fi
    python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels_ms_per_frame']
print('$v', '| ms/frame', d['ms_per_step'], '| full path', d['config'].get('full_path_ms_per_frame'), '|', {n:v for n,v in k.items() if 'conv48' in n or 'stem' in n or n=='gemm'}, '| sum', round(sum(k.values()),3))"
  done
done
