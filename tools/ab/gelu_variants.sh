# A/B of the GELU on packed (v_pk_*_f32) against single-value fp32 instructions in the two MLP kernels (run on the GPU box)
set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans"
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; (cd ..; python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"); }
run base
$CXX -DW2X_GELU_SCALAR -fno-slp-vectorize -c csrc/k_mlp96p.hip -o build/k_mlp96p.o
run mlp96p_scalar
$CXX -DW2X_GELU_SCALAR -fno-slp-vectorize -c csrc/k_mlp2.hip -o build/k_mlp2.o
run both_scalar
$CXX -DW2X_GELU_SCALAR -c csrc/k_mlp96p.hip -o build/k_mlp96p.o
$CXX -DW2X_GELU_SCALAR -c csrc/k_mlp2.hip -o build/k_mlp2.o
run both_scalar_slp
