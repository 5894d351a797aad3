# A/B of two builds of k_swinattn192 on one box: the shipped source against tools/ab/k_swinattn192_r2a.hip (a wave = (window, 3 heads), two alternating weight register sets)
set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -I csrc"
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; (cd ..; python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"); }
run shipped
$CXX -c ../tools/ab/k_swinattn192_r2a.hip -o build/k_swinattn192.o
run r2a
$CXX -c csrc/k_swinattn192.hip -o build/k_swinattn192.o
run shipped
$CXX -c ../tools/ab/k_swinattn192_r2a.hip -o build/k_swinattn192.o
run r2a
