for i in 1 2; do
for cfg in "W2X_A192_TWO_PER_CU=1 W2X_NO_SPLIT=1" "W2X_NO_SPLIT=1" "W2X_A192_TWO_PER_CU=1" "X=1" "W2X_GROUPS=3"; do
  env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['config']['full_path_ms_per_frame'])"
done; done
