# A/B of the GELU polynomial length in the two MLP kernels (run on the GPU box): time and network parity
set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../include -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans"
for d in 6 5 4; do
  $CXX -DW2X_GELU_DEG=$d -c csrc/k_mlp96p.hip -o build/k_mlp96p.o
  $CXX -DW2X_GELU_DEG=$d -c csrc/k_mlp2.hip -o build/k_mlp2.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
  (cd ..; python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('deg$d',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"
   rm -f gpurun_out/parity/parity.jsonl; python -m pytest tests -m gpu -x -q -k "network_matches and swin" 2>&1 | tail -1
   python -c "
import json
r=[json.loads(l) for l in open('gpurun_out/parity/parity.jsonl') if 'network' in l]
print('deg$d max_ulp16', max(x['max_ulp16'] for x in r), 'mean_abs', max(x['mean_abs'] for x in r))")
done
