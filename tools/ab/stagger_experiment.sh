mkdir -p gpurun_out/r3_stagger
for k in -1 0 1 2 3 5 -1; do
  W2X_STAGGER_OP=$k python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stagger $k', d['ms_per_step'], d['config']['full_path_ms_per_frame'])"
done | tee gpurun_out/r3_stagger/stagger.txt
