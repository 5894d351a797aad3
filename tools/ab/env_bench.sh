#!/bin/bash
# bench.py under a list of environment settings, alternating, one line each: resident frame / host-to-host sequence / synchronous render() call.
# usage (GPU box): bash tools/ab/env_bench.sh ROUNDS "" "NAME=VALUE" "NAME=VALUE NAME2=VALUE2" ...
cd "$(dirname "$0")/../.." || exit 1
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for setting in "$@"; do
    env $setting python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('[$setting]', 'resident', d['ms_per_step'], d.get('ms_per_step_samples'), '| sequence', d['config']['full_path_ms_per_frame'], '| render()', d['config']['render_call_ms_pageable'])"
  done
done
