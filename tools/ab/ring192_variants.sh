set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -I csrc"
run() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o; (cd ..; python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['value'],d['ms_per_step'],d['roofline']['kernels_ms_per_frame'])"); }
for r in 8 12 6 8 12; do
  sed "s/constexpr int RING = 8, NFRAG = 72;/constexpr int RING = $r, NFRAG = 72;/" csrc/k_swinattn192.hip > /tmp/k192_r$r.hip
  $CXX -c /tmp/k192_r$r.hip -o build/k_swinattn192.o 2>/dev/null
  run ring$r
done
