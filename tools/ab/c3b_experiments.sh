# conv3c_kernel: where does the time go?  Variants: 1 weight fetches hit one address (L1 hits), 2 no halo fetch, 3 no stores, 4 no products
set -u
cd $GRAFT_REPO_ROOT/waifu2x-tensorrt_amd
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../include -Wno-unused-function -Wno-unused-variable"
for v in 0 1 2 3 4; do
  $CXX -DW2X_C3B_EXP=$v -c csrc/k_conv3.hip -o build/k_conv3.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
  (cd ..; echo "== variant $v"; python tools/op_times.py cunet/art 2 1 4 256 1080 1920 2>&1 | grep -E "ms per resident|conv3x3s1.*(K=288|K=576|K=1152|K=2304) N=(64|128|256)" | cut -c1-90)
done
