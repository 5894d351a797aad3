#!/bin/bash
# Tile groups per pass (W2X_GROUPS) re-measured on the GPU box:  tools/ab/groups_experiment.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r3_groups
for g in 2 1 3 4 2; do
  W2X_GROUPS=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('groups $g', d['ms_per_step'], d['config']['full_path_ms_per_frame'])"
done | tee gpurun_out/r3_groups/groups.txt
