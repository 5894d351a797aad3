mkdir -p gpurun_out/r3_head
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export W2X_NO_FUSE_HEAD=1; else unset W2X_NO_FUSE_HEAD; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no_fuse=$v', d['ms_per_step'], d['config']['full_path_ms_per_frame'], d['roofline']['kernels_ms_per_frame'])"
done | tee gpurun_out/r3_head/ab.txt
