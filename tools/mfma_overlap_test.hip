// Hardware check: may the destination of v_mfma_f32_16x16x32_f16 overlap its A or B operand registers?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_overlap_test.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(const half8* A, const half8* B, const float4v* Cin, float4v* out) {
    const int lane = threadIdx.x;
    half8 a = A[lane], b = B[lane];
    float4v c = Cin[lane];
    float4v d0, d1, d2;
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "=&v"(d0) : "v"(a), "v"(b), "v"(c));
    float4v xa = __builtin_bit_cast(float4v, a);
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 %0, %0, %1, %2\n\ts_nop 15\n\ts_nop 15" : "+v"(xa) : "v"(b), "v"(c));
    d1 = xa;
    float4v xb = __builtin_bit_cast(float4v, b);
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %0, %2\n\ts_nop 15\n\ts_nop 15" : "+v"(xb) : "v"(a), "v"(c));
    d2 = xb;
    // destination shifted by two registers against srcC (partial overlap), as hipcc emitted in k_mlp2.hip
    float4v d3;
    asm volatile("v_mov_b32 v66, %5\n\tv_mov_b32 v67, %6\n\tv_mov_b32 v68, %7\n\tv_mov_b32 v69, %8\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_f16 v[68:71], %9, %10, v[66:69]\n\ts_nop 15\n\ts_nop 15\n\t"
                 "v_mov_b32 %0, v68\n\tv_mov_b32 %1, v69\n\tv_mov_b32 %2, v70\n\tv_mov_b32 %3, v71\n\tv_mov_b32 %4, v66"
                 : "=&v"(d3[0]), "=&v"(d3[1]), "=&v"(d3[2]), "=&v"(d3[3]), "=&v"(d1[0])
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b)
                 : "v66", "v67", "v68", "v69", "v70", "v71");
    d1 = xa;
    out[lane] = d0; out[64 + lane] = d1; out[128 + lane] = d2; out[192 + lane] = d3;
}
int main() {
    half8 hA[64], hB[64]; float4v hC[64], hO[256];
    srand(1);
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 8; ++e) { hA[i][e] = (_Float16)((rand() % 200 - 100) / 50.f); hB[i][e] = (_Float16)((rand() % 200 - 100) / 50.f); }
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) hC[i][e] = (rand() % 100) / 10.f;
    half8 *dA, *dB; float4v *dC, *dO;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dO, sizeof(hO));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
    hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) { e1 = fmax(e1, fabs(hO[i][e] - hO[64 + i][e])); e2 = fmax(e2, fabs(hO[i][e] - hO[128 + i][e])); e3 = fmax(e3, fabs(hO[i][e] - hO[192 + i][e])); }
    printf("dst overlaps A: max diff %.5f   dst overlaps B: max diff %.5f   dst = srcC shifted by 2: max diff %.5f   (sample d0=%.3f)\n", e1, e2, e3, hO[5][1]);
    return 0;
}
