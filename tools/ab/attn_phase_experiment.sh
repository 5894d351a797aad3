# Timing experiment (wrong results): the attention kernels with their row fetches served from one 16 KB region (cache hits) and / or
# their stores dropped - how much of a launch is exposed memory latency?  Needs the W2X_ATTN_EXP hooks (bit 0: fetch offsets
# masked to 0x3FF0, bit 1: store offsets replaced by the out-of-range offset) patched into the two kernels' buffer loads / stores;
# result of the round-2 run: profiles/r2_final/attn_phase_experiment.txt (C=96: -12.5 % / -10 % / -13.7 %, C=192: -5 % / -5 % / -10 %).
set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../include -Wno-unused-function -Wno-unused-variable"
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; (cd ..; python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"); }
for v in 0 1 2 3; do
  $CXX -DW2X_ATTN_EXP=$v -c csrc/k_swinattn96.hip -o build/k_swinattn96.o
  $CXX -DW2X_ATTN_EXP=$v -c csrc/k_swinattn192.hip -o build/k_swinattn192.o
  run exp$v
done
