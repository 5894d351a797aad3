#!/bin/bash
# Timing experiments on conv48p_kernel (k_conv48p.hip W2X_C48P_EXP): which phase of a tile the time goes to.  Prints the op's HIP-event time per variant.
cd "$(dirname "$0")/../../waifu2x-tensorrt_amd" || exit 1
CXX="/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
for v in ${VARIANTS:-0 1 2 4 8 6 7 15}; do
  $CXX -DW2X_C48P_EXP=$v -I csrc -c csrc/k_conv48p.hip -o build/k_conv48p.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
  echo -n "EXP=$v: "; (cd ..; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --op-times 2>&1 | grep -E "^ +[0-9.]+ ms +1 gemm" | cut -c1-14)
done
$CXX -I csrc -c csrc/k_conv48p.hip -o build/k_conv48p.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libw2x.so build/*.o
