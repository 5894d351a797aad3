// What the matrix pipe sustains at the board's power cap: back-to-back v_mfma on registers (no memory, no LDS), every SIMD with two or four waves, for a few seconds.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_cap tools/ab/mfma_cap.hip && python tools/power_trace.py -- /tmp/mfma_cap
// Prints TFLOP/s per instruction shape; tools/power_trace.py beside it gives the clock and the watts the loop ran at.  Operand values: small non-zero halves
// (a pipe multiplying zeros draws much less: an ablation with wrong data measures the cap).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void burn(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed * (float)((lane * 7 + j * 3) % 11 - 5) * 0.01f); b[j] = (_Float16)(seed * (float)((lane * 5 + j) % 13 - 6) * 0.01f); }
    float sum = 0.f;
    if (SHAPE == 16) {
        float4v acc[8];
        for (int u = 0; u < 8; ++u) acc[u] = (float4v){seed * u, seed * (u + lane), 1.f + u, 2.f * u};       // (distinct chains: identical ones are merged by the compiler)
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
        }
        for (int u = 0; u < 8; ++u) sum += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    } else {
        float16v acc[4];
        for (int u = 0; u < 4; ++u) for (int j = 0; j < 16; ++j) acc[u][j] = seed * (u * 16 + j) + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) for (int j = 0; j < 16; ++j) sum += acc[u][j];
    }
    if (sum == 12345.678f) out[0] = sum;      // keeps the loop
}

template <int SHAPE>
void run(const char* name, int wgs_per_cu, double seconds) {
    float* out; hipMalloc(&out, 4);
    const int cus = 256, iters = 20000;
    const double flop_per_launch = (double)cus * wgs_per_cu * 4 /*waves*/ * iters * (SHAPE == 16 ? 8 * 16384.0 : 4 * 32768.0);
    burn<SHAPE><<<cus * wgs_per_cu, 256>>>(out, 100, 1.f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0, total_ms = 0; int n = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0); burn<SHAPE><<<cus * wgs_per_cu, 256>>>(out, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; ++n;
        best = std::max(best, flop_per_launch / (ms * 1e-3) * 1e-12);
    }
    printf("%s, %d waves per SIMD: mean %.0f TFLOP/s over %d launches (%.1f s), best launch %.0f\n", name, wgs_per_cu, flop_per_launch * n / (total_ms * 1e-3) * 1e-12, n, total_ms * 1e-3, best);
    hipFree(out);
}
int main(int argc, char** argv) {      // argument: 16 or 32 (one shape per process, so that tools/power_trace.py shows that shape's clock and watts); none: both
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    if (only != 32) { run<16>("v_mfma_f32_16x16x32_f16", 2, 4.0); run<16>("v_mfma_f32_16x16x32_f16", 4, 4.0); }
    if (only != 16) { run<32>("v_mfma_f32_32x32x16_f16", 2, 4.0); run<32>("v_mfma_f32_32x32x16_f16", 4, 4.0); }
    return 0;
}
