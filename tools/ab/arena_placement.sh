#!/bin/bash
# Placement experiment on the GPU box: per-op times of one frame (bench.py --op-times) with the activation arena's tensors aligned to
# different units (W2X_ARENA_ALIGN, engine.cpp upload_plan).  Same kernels, same sizes, different addresses - the question it answers is
# whether the 5-10 % spread between launches of one kernel at one size (encoder vs decoder blocks) follows the addresses.
#   tools/ab/arena_placement.sh <out dir> [align ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$1; shift
mkdir -p "$OUT"
for al in "${@:-3072}"; do
  W2X_DUMP_ARENA=1 W2X_ARENA_ALIGN=$al python bench.py --steps 10 --warmup 2 --no-cpu-baseline --op-times > "$OUT/align_$al.json" 2> "$OUT/align_$al.txt"
  echo "align $al: $(python -c "import json;print(json.load(open('$OUT/align_$al.json'))['ms_per_step'])") ms/frame | $(grep ' ms ' "$OUT/align_$al.txt" | awk '{printf "%s ", $1}')"
done
