# A/B of -fno-honor-nans on the remaining kernel files (run on the GPU box): rebuilds single objects, relinks, runs bench.py
set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; (cd ..; python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"); }
$CXX -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans -c csrc/k_mlp96p.hip -o build/k_mlp96p.o
run base
$CXX -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans -c csrc/k_mlp2.hip -o build/k_mlp2.o
run mlp2_nonan
$CXX -fno-honor-nans -c csrc/k_swinattn192.hip -o build/k_swinattn192.o
run attn192_nonan
$CXX -fno-honor-nans -c csrc/k_swinattn96.hip -o build/k_swinattn96.o
run attn96_nonan
