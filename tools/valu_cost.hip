// Cost of single vector instructions on gfx950 in the unit tools/issue_model.hip uses (s_memtime ticks per instruction and SIMD with three waves per SIMD
// issuing streams of INDEPENDENT instructions of one kind): which of the instructions the transformer kernels are made of take more than one issue slot?
//   hipcc --offload-arch=gfx950 -O2 tools/valu_cost.hip -o tools/valu_cost && tools/valu_cost        (GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 4000, N = 16;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc) {
    float v[8]; float2v p[8]; unsigned u[8];
    for (int j = 0; j < 8; ++j) { v[j] = 1.f + j + threadIdx.x * 1e-3f; p[j] = (float2v){v[j], v[j] + 0.5f}; u[j] = j * 77u + threadIdx.x; }
    const float m = 0.999f, ad = 0.001f; const unsigned long long smask = 0x5555AAAA5555AAAAull + blockIdx.x; const float sm = 0.999f + blockIdx.x * 1e-9f; unsigned long long sm2 = smask; const float2v m2 = {0.999f, 0.998f}, a2 = {0.001f, 0.002f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int r = j & 7;
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(m), "v"(ad));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(m2), "v"(a2));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(m2));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(a2));
            if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
            if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(v[r]), "v"(v[(r + 1) & 7]));
            if (KIND == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(m), "v"(ad));
            if (KIND == 7) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[r]) : "v"(u[r]), "v"(u[(r + 1) & 7]));
            if (KIND == 8) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(m), "v"(ad));
            if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
            if (KIND == 10) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[r]));
            if (KIND == 11) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
            if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
            if (KIND == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
            if (KIND == 14) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
            if (KIND == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[r]) : "v"(m));
            if (KIND == 16) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));          // no chain through the destination
            if (KIND == 17) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]), "s"(smask));
            if (KIND == 18) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 2) & 7]), "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
            if (KIND == 19) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[r]), "v"(m) : "vcc");
            if (KIND == 20) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(u[r]) : "v"(v[r]), "v"(m), "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]) : "vcc");
            if (KIND == 31 && (j & 3) == 0) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[r]), "v"(m) : "vcc");                         // one compare, three selects on its vcc
            if (KIND == 31 && (j & 3) != 0) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
            if (KIND == 32 && (j & 7) == 0) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[r]), "v"(m) : "vcc");                         // one compare, seven selects
            if (KIND == 32 && (j & 7) != 0) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
            if (KIND == 33 && (j & 7) == 0) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(sm2) : "v"(v[r]), "v"(m));                        // one compare into an SGPR pair, seven selects
            if (KIND == 33 && (j & 7) != 0) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]), "s"(sm2));
            if (KIND == 34) asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n\tv_mul_f32 %3, %3, %4" : "=v"(u[r]), "+v"(v[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]), "v"(m));   // select + multiply alternating
            if (KIND == 21) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[r]) : "v"(u[(r + 3) & 7]));
            if (KIND == 22) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(v[r]) : "v"(u[(r + 3) & 7]), "v"(m), "v"(ad));
            if (KIND == 23) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
            if (KIND == 24) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 2) & 7]), "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
            if (KIND == 25) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[(r + 3) & 7]), "v"(m));
            if (KIND == 26) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[r]) : "v"(v[(r + 3) & 7]), "v"(m), "v"(ad));                     // no chain
            if (KIND == 27) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[(r + 3) & 7]), "v"(v[(r + 5) & 7]));
            if (KIND == 28) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 2) & 7]), "v"(u[(r + 3) & 7]), "v"(u[(r + 5) & 7]));
            if (KIND == 29) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[(r + 3) & 7]), "s"(sm));                                     // one SGPR operand
            if (KIND == 30) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[r]) : "v"(v[(r + 3) & 7]), "v"(m));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)(sm2 & 1);
    for (int j = 0; j < 8; ++j) s += v[j] + p[j][0] + p[j][1] + (float)u[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc, int ncu) {
    for (int W : {1, 3}) {
        const int blocks = ncu * W;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc); CK(hipDeviceSynchronize()); }
        std::vector<long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("%-18s W=%d: %.2f ticks per instruction and SIMD\n", name, W, (double)h[h.size() / 2] / ITER / N / W);
    }
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    float* out; long long* cyc;
    CK(hipMalloc(&out, (size_t)ncu * 4 * 256 * 4)); CK(hipMalloc(&cyc, (size_t)ncu * 4 * 4 * 8));
    run<0>("v_fma_f32", out, cyc, ncu); run<15>("v_mul_f32", out, cyc, ncu); run<1>("v_pk_fma_f32", out, cyc, ncu); run<2>("v_pk_mul_f32", out, cyc, ncu); run<3>("v_pk_add_f32", out, cyc, ncu);
    run<4>("v_exp_f32", out, cyc, ncu); run<13>("v_rcp_f32", out, cyc, ncu); run<5>("v_cvt_pk_f16_f32", out, cyc, ncu); run<6>("v_max3_f32", out, cyc, ncu);
    run<7>("v_dot2c_f32_f16", out, cyc, ncu); run<8>("v_fma_mixlo_f16", out, cyc, ncu); run<9>("v_mov_b32", out, cyc, ncu); run<10>("v_add_f32_dpp", out, cyc, ncu);
    run<16>("v_cndmask vcc free", out, cyc, ncu); run<17>("v_cndmask sgpr", out, cyc, ncu); run<18>("v_bfi_b32", out, cyc, ncu); run<19>("v_cmp_gt_f32", out, cyc, ncu); run<20>("v_cmp + v_cndmask", out, cyc, ncu);
    run<21>("v_cvt_f32_f16", out, cyc, ncu); run<22>("v_fma_mix_f32", out, cyc, ncu); run<23>("v_pack_b32_f16", out, cyc, ncu); run<24>("v_perm_b32", out, cyc, ncu); run<25>("v_max_f32", out, cyc, ncu);
    run<26>("v_fma_f32 free", out, cyc, ncu); run<27>("v_add_f32 vv", out, cyc, ncu); run<28>("v_pk_fma_f16", out, cyc, ncu); run<29>("v_mul_f32 sgpr", out, cyc, ncu); run<30>("v_fmac_f32", out, cyc, ncu);
    run<31>("cmp + 3 cndmask vcc", out, cyc, ncu); run<32>("cmp + 7 cndmask vcc", out, cyc, ncu); run<33>("cmp(s) + 7 cndmask s", out, cyc, ncu); run<34>("cndmask vcc + mul", out, cyc, ncu);
    run<11>("v_pk_add_f16", out, cyc, ncu); run<14>("v_pk_mul_f16", out, cyc, ncu); run<12>("v_cndmask_b32", out, cyc, ncu);
    return 0;
}
