#!/bin/bash
# The library of the round's first commit against the library as built now, alternating on ONE box: frame time (median of seven regions) and, from tools/power_trace.py around a
# 300-frame run, watts and clock - i.e. joules per frame.  tools/ab/libw2x_round_start.so is built beforehand from `git archive <first commit> waifu2x-tensorrt_amd include`
# (make libw2x.so) and is not tracked.   GPU box:  bash tools/ab/round_over_round.sh [config] [rounds]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
CFG=${1:-3}; ROUNDS=${2:-2}
cp waifu2x-tensorrt_amd/libw2x.so /tmp/libw2x_now.so
one() {   # <label> <library>
  cp "$2" waifu2x-tensorrt_amd/libw2x.so
  python bench.py --config "$CFG" --no-cpu-baseline --work "/tmp/w2x_bench_$1" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/frame', d['ms_per_step'], d['ms_per_step_samples'], 'host to host', d['host_to_host']['ms_per_frame'])"
  python tools/power_trace.py -- python bench.py --config "$CFG" --steps 300 --repeats 1 --no-cpu-baseline --work "/tmp/w2x_bench_$1" 2>/dev/null | python -c "
import json,sys
ms=None
for l in sys.stdin:
    if l.startswith('{'): ms=json.loads(l)['ms_per_step']
    if l.startswith('POWER'):
        p=json.loads(l[6:]); print('$1', '   300 frames:', ms, 'ms/frame at', p['busy_power_mean_w'], 'W mean,', p['busy_sclk_mean_mhz'], 'MHz ->', round(ms*1e-3*p['busy_power_mean_w'],2), 'J per frame')"
}
for r in $(seq 1 "$ROUNDS"); do
  one round_start tools/ab/libw2x_round_start.so
  one now /tmp/libw2x_now.so
done
cp /tmp/libw2x_now.so waifu2x-tensorrt_amd/libw2x.so
