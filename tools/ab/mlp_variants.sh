set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
for v in "maxilp_nonan:-mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans" "default_nonan:-fno-honor-nans" "maxilp:-mllvm -amdgpu-sched-strategy=max-ilp"; do
  name=${v%%:*}; fl=${v#*:}
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable $fl -c csrc/k_mlp96p.hip -o build/k_mlp96p.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
  cd ..; python bench.py --no-cpu-baseline --steps 10 --op-times 2> /tmp/err.txt | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"; grep -E " mlp C=96" /tmp/err.txt | head -2; cd waifu2x-tensorrt_amd
done
