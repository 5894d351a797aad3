// Issue-model calibration for the VALU-heavy transformer kernels (GPU box only):
//   hipcc --offload-arch=gfx950 -O2 tools/issue_model.hip -o tools/issue_model && tools/issue_model
// Question: with W waves per SIMD each running a stream of (one MFMA + NV independent VALU instructions), how many SIMD cycles does
// one such group cost?  MI355X_MICROARCH.md gives the one-wave figures (a VALU instruction 4 issue cycles, an MFMA holds the vector
// issue for 8 of its 16 / 32 cycles, costs add); the kernels here run 3-4 waves per SIMD, and the choice between
// v_mfma_f32_16x16x32_f16 (16 cycles, 16 K FLOP) and v_mfma_f32_32x32x16_f16 (32 cycles, 32 K FLOP: half the MFMA instructions for
// the same product) depends on whether the 8-cycle hold is per instruction in that regime too.
// Every wave runs ITER groups between two s_memtime stamps; printed: SIMD cycles per group = (t1 - t0) / ITER / W (the W waves of
// a SIMD run concurrently), and the same per 16 K FLOP of matrix work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;

// SHAPE 0: 16x16x32, 1: 32x32x16, 2: 16x16x16, 3: no MFMA, 4 / 5: 16x16x32 / 32x32x16 with the accumulators in AccVGPRs (does the matrix
// instruction then leave the vector register file's ports to the VALU work of the other waves?).  NV: plain v_fma_f32 per group.  NX: v_exp_f32 per group.
template <int SHAPE, int NV, int NX>
__global__ __launch_bounds__(256) void k(const half8* in, float* out, long long* cyc) {
    const int lane = threadIdx.x & 63;
    half8 a = in[lane], b = in[64 + lane];
    half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    float4v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float16v d0 = {}, d1 = {};
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    float x[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    const float m = 0.999f, ad = 0.001f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SHAPE == 0) {
                float4v& c = u == 0 ? c0 : u == 1 ? c1 : u == 2 ? c2 : c3;
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
            } else if (SHAPE == 1) {
                float16v& d = (u & 1) ? d1 : d0;
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
            } else if (SHAPE == 2) {
                float4v& c = u == 0 ? c0 : u == 1 ? c1 : u == 2 ? c2 : c3;
                asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a4), "v"(b4));
            } else if (SHAPE == 4) {
                float4v& c = u == 0 ? c0 : u == 1 ? c1 : u == 2 ? c2 : c3;
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
            } else if (SHAPE == 5) {
                float16v& d = (u & 1) ? d1 : d0;
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(m), "v"(ad));
#pragma unroll
            for (int j = 0; j < NX; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 3]));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int NV, int NX>
void run(const char* name, const half8* in, float* out, long long* cyc, int ncu) {
    for (int W : {1, 2, 3, 4}) {
        const int blocks = ncu * W;
        // W workgroups of four waves per CU: one wave of each per SIMD (the kernels' own geometry)
        hipLaunchKernelGGL((k<SHAPE, NV, NX>), dim3(blocks), dim3(256), 0, 0, in, out, cyc);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((k<SHAPE, NV, NX>), dim3(blocks), dim3(256), 0, 0, in, out, cyc);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2] / ITER;          // wave cycles per group
        const double flopk = (SHAPE == 1 || SHAPE == 5) ? 2.0 : SHAPE == 2 ? 0.5 : (SHAPE == 0 || SHAPE == 4) ? 1.0 : 0.0;   // 16 K FLOP units per MFMA
        printf("%-12s NV=%d NX=%d W=%d: %.1f wave cycles per group -> %.1f SIMD cycles per group", name, NV, NX, W, med, med / W);
        if (flopk > 0) printf(", %.1f per 16 KFLOP", med / W / flopk);
        printf("\n");
    }
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    std::vector<_Float16> hin(128 * 8);
    for (auto& v : hin) v = (_Float16)((rand() % 200 - 100) / 400.f);
    half8* in; float* out; long long* cyc;
    CK(hipMalloc(&in, hin.size() * 2)); CK(hipMemcpy(in, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)ncu * 4 * 256 * 4)); CK(hipMalloc(&cyc, (size_t)ncu * 4 * 4 * 8));
    printf("CUs %d, %d groups per wave; s_memtime ticks (shader cycles)\n", ncu, ITER);
    run<3, 8, 0>("valu only", in, out, cyc, ncu);
    run<3, 0, 4>("exp only", in, out, cyc, ncu);
    run<0, 0, 0>("16x16x32", in, out, cyc, ncu);
    run<1, 0, 0>("32x32x16", in, out, cyc, ncu);
    run<2, 0, 0>("16x16x16", in, out, cyc, ncu);
    run<0, 2, 0>("16x16x32", in, out, cyc, ncu);
    run<0, 4, 0>("16x16x32", in, out, cyc, ncu);
    run<0, 6, 0>("16x16x32", in, out, cyc, ncu);
    run<0, 8, 0>("16x16x32", in, out, cyc, ncu);
    run<1, 4, 0>("32x32x16", in, out, cyc, ncu);
    run<1, 8, 0>("32x32x16", in, out, cyc, ncu);
    run<1, 12, 0>("32x32x16", in, out, cyc, ncu);
    run<1, 16, 0>("32x32x16", in, out, cyc, ncu);
    run<0, 4, 1>("16x16x32", in, out, cyc, ncu);
    run<1, 8, 2>("32x32x16", in, out, cyc, ncu);
    run<2, 4, 0>("16x16x16", in, out, cyc, ncu);
    run<4, 0, 0>("16x16x32 acc", in, out, cyc, ncu);
    run<4, 4, 0>("16x16x32 acc", in, out, cyc, ncu);
    run<4, 8, 0>("16x16x32 acc", in, out, cyc, ncu);
    run<5, 0, 0>("32x32x16 acc", in, out, cyc, ncu);
    run<5, 8, 0>("32x32x16 acc", in, out, cyc, ncu);
    run<5, 16, 0>("32x32x16 acc", in, out, cyc, ncu);
    return 0;
}
