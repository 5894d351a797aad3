// Fused transformer MLP branch, C = 96, resident-weight variant for gfx950:   y = x + W2 * gelu(W1 * LayerNorm(x) + b1) + b2
// Same math, parameters and per-wave dataflow as k_mlp2.hip (a wave owns 32 token rows from the first load to the last store,
// GEMM1 and GEMM2 both transposed, GELU'd accumulators used as the next operand as they stand), but at this width both weight
// matrices together are 72 KiB - they fit the CU's LDS next to the waves' row slabs.  So the kernel is persistent: one workgroup of
// 12 waves per CU copies the fragment-major weights into LDS once, and every wave then walks its own sequence of 32-row tiles
// with no workgroup barrier and no weight traffic at all (k_mlp2.hip re-stages the 72 KiB for every 128 rows - more bytes than
// the rows themselves - behind one barrier per 32-hidden-unit chunk; its waves spent 44 % of their time parked).
// Three waves per SIMD, each in its own phase, cover each other's memory latency.
#include "kernels.h"
#ifndef W2X_GELU_DEG
#define W2X_GELU_DEG 4   // coefficients of q(u): 6 -> 3.1e-7, 5 -> 7.1e-7, 4 -> 8.7e-6 absolute error of GELU (tools/fit_gelu.py).  4: a third of
                         // the fp16 rounding of the smallest hidden values that matter, network parity unchanged (2.0 ULP16 on every full-width
                         // graph, same mean error), MLP kernels 5-7 % faster (round 2, profiles/r2_final/gelu_degree_ab.txt; now: tools/ab/lib_variants.sh "k_mlp2.hip:-DW2X_GELU_DEG=6")
#endif

#include <algorithm>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// GELU(x) = max(x,0) - 0.5 u 2^-q(u), u = min(|x|, 6.5): tools/fit_gelu.py (|err| < 8.7e-6 with the four-coefficient q), two values at a time on v_pk_*_f32
__device__ __forceinline__ float2v splat2(float c) { return (float2v){c, c}; }
#ifdef W2X_GELU_SCALAR   // A/B: the same polynomial on single-value instructions
__device__ __forceinline__ float gelu_fast1(float x) {
    const float u = fminf(fabsf(x), 6.5f);
    float q = fmaf(-2.992485764e-05f, u, 7.398797018e-04f);
    q = fmaf(q, u, -7.977479093e-03f);
    q = fmaf(q, u, 5.323820859e-02f);
    q = fmaf(q, u, 4.589156733e-01f);
    q = fmaf(q, u, 1.151147085e+00f);
    return fmaf(-0.5f * u, __builtin_amdgcn_exp2f(-(q * u)), fmaxf(x, 0.f));
}
__device__ __forceinline__ float2v gelu_fast2(float2v x) { return (float2v){gelu_fast1(x[0]), gelu_fast1(x[1])}; }
#else
__device__ __forceinline__ float2v gelu_fast2(float2v x) {
    const float2v u = {fminf(fabsf(x[0]), 6.5f), fminf(fabsf(x[1]), 6.5f)};
#if W2X_GELU_DEG == 5
    float2v q = __builtin_elementwise_fma(splat2(4.881020589e-04f), u, splat2(-7.198718011e-03f));
    q = __builtin_elementwise_fma(q, u, splat2(5.214663110e-02f));
    q = __builtin_elementwise_fma(q, u, splat2(4.595958449e-01f));
    q = __builtin_elementwise_fma(q, u, splat2(1.151000542e+00f));
#elif W2X_GELU_DEG == 4
    float2v q = __builtin_elementwise_fma(splat2(-4.161669730e-03f), u, splat2(4.573546095e-02f));
    q = __builtin_elementwise_fma(q, u, splat2(4.649304537e-01f));
    q = __builtin_elementwise_fma(q, u, splat2(1.149566979e+00f));
#else
    float2v q = __builtin_elementwise_fma(splat2(-2.992485764e-05f), u, splat2(7.398797018e-04f));
    q = __builtin_elementwise_fma(q, u, splat2(-7.977479093e-03f));
    q = __builtin_elementwise_fma(q, u, splat2(5.323820859e-02f));
    q = __builtin_elementwise_fma(q, u, splat2(4.589156733e-01f));
    q = __builtin_elementwise_fma(q, u, splat2(1.151147085e+00f));
#endif
    const float2v t = __builtin_elementwise_fma(q, u, splat2(1.f));              // the factor 1/2 rides in the exponent: 0.5 * 2^-qu = 2^-(qu + 1)
    const float2v e = {__builtin_amdgcn_exp2f(-t[0]), __builtin_amdgcn_exp2f(-t[1])};
    const float2v m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-u, e, m);
}
#endif
__device__ __forceinline__ void sum_sq8(const half8 v, float& s, float& q) {
    const half2v one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const half2v h = {v[2 * k], v[2 * k + 1]};
        s = __builtin_amdgcn_fdot2(h, one, s, false);
        q = __builtin_amdgcn_fdot2(h, h, q, false);
    }
}
// sums over the four 16-lane rows of a wave for two values at once: the two chains fill each other's permlane wait states.
// The inputs come straight from v_dot2c chains: a dot result needs 3 wait states before a different VALU may read it, and
// nothing inside an asm statement is padded by the compiler - hence the leading s_nop 2 (without it the sums were wrong on some
// waves of some launches).
__device__ __forceinline__ void rows_sum2(float& a0, float& a1) {
    float b0, b1;
    asm volatile(
        "s_nop 2\n\tv_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\t"
        "v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\t"
        "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"
        "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
        "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3"
        : "+v"(a0), "+v"(a1), "=&v"(b0), "=&v"(b1));
}
// (x * rstd + nm) on 8 halves with fp32 arithmetic: v_fma_mixlo / mixhi read the f16 halves directly and write f16
__device__ __forceinline__ half8 norm8(const half8 v, float rstd, float nm) {
    uint4v x = __builtin_bit_cast(uint4v, v), o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned r;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x[d]), "v"(rstd), "v"(nm));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(x[d]), "v"(rstd), "v"(nm));
        o[d] = r;
    }
    return __builtin_bit_cast(half8, o);
}

// A wave hands data from lane to lane through its own LDS slab; the hardware executes a wave's LDS instructions in order, so a
// compiler-level fence (no instruction) is all that is needed between the phases.
#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")
#define W2X_RING_FENCE() asm volatile("" ::: "memory")   // keeps a ring refill where it is written (the scheduler would sink it to its use)

constexpr int C = 96, TT = 2, RW = 16 * TT, NWV = 12, NTHR = NWV * 64;
constexpr int LDX = C + 8, PPR = C / 8, KS = C / 32, NT = C / 16, NCH = 2 * C / 32;
constexpr int NP = RW * PPR / 64;            // flat 16-byte pieces per lane (6)
constexpr int SLAB = RW * LDX * 2;           // bytes per wave
constexpr int W1F = NCH * 2 * KS, W2F = NCH * NT;   // KiB fragments of the two matrices (36 + 36)
constexpr int NF = 2 * KS;                   // fragments per chunk and matrix = ring registers (6)
static_assert(NF == NT, "one ring serves both products");
constexpr int WBYTES = (W1F + W2F) * 1024;
constexpr int SMEM96P = WBYTES + NWV * SLAB;
static_assert(RW * PPR % 64 == 0, "flat piece count");
static_assert(SMEM96P <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(NTHR, 3) void mlp96p_kernel(const MlpParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const _Float16* WL = (const _Float16*)smem + lane * 8;                       // weights: [W1F + W2F fragments][64 lanes][8]
    _Float16* Xw = (_Float16*)(smem + WBYTES + wv * SLAB);                      // this wave's slab [RW][LDX]

    // ---- both matrices, fragment-major as engine.cpp stores them, into LDS - once per workgroup
    {
        const uint4v* w1 = (const uint4v*)p.w1_frag;
        const uint4v* w2 = (const uint4v*)p.w2_frag;
        uint4v* dst = (uint4v*)smem;
        constexpr int N1 = W1F * 64, NALL = (W1F + W2F) * 64;
#pragma unroll
        for (int k = 0; k < NALL / NTHR; ++k) {
            const int i = k * NTHR + tid;
            dst[i] = i < N1 ? w1[i] : w2[i - N1];
        }
    }
    __syncthreads();

    // Weight fragments reach the MFMAs through a ring of six registers: a fragment is requested from LDS right after the last MFMA
    // that used its register - W2's during GEMM1 (they land under the GELU), the next chunk's W1's during GEMM2 (they land under the
    // rest of GEMM2) - so no product waits on an LDS round trip.  (Reads placed at their point of use made the compiler wait for
    // each one: 72 exposed LDS latencies per tile.)
    half8 wr[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) wr[i] = *(const half8*)(WL + (size_t)((i & 1) * KS + (i >> 1)) * 512);
    const int gw = blockIdx.x * NWV + wv, nw = gridDim.x * NWV;
#pragma unroll 1
    for (int tile = gw; tile < ntiles; tile += nw) {
        const long row0 = (long)tile * RW;
        const long nrows = p.M - row0 < RW ? p.M - row0 : RW;
        const int npieces = (int)nrows * PPR;
        const _Float16* __restrict__ X = (const _Float16*)p.x + row0 * C;
        // ---- x rows: flat coalesced load -> slab (the raw rows stay there for the residual add)
        {
            half8 xr[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int idx = k * 64 + lane;
                half8 h = {};
                if (idx < npieces) h = *(const half8*)(X + (size_t)idx * 8);
                xr[k] = h;
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
                *(half8*)(Xw + r * LDX + c * 8) = xr[k];
            }
        }
        W2X_PHASE_FENCE();
        // ---- LayerNorm in fragment layout: lane (fr, g) holds channels ks*32 + 8g .. +7 of row 16tt + fr, so the row sums are the
        //      lane's own KS pieces plus the three other lane groups; the normalised pieces are the operand registers of GEMM1
        half8 xreg[TT][KS];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            half8 raw[KS];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { raw[ks] = *(const half8*)(Xw + (tt * 16 + fr) * LDX + ks * 32 + g * 8); sum_sq8(raw[ks], s, q); }
            rows_sum2(s, q);
            const float mean = s * (1.f / C);
            const float rstd = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps);
            const float nm = -mean * rstd;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xreg[tt][ks] = norm8(raw[ks], rstd, nm);
        }
        // GEMM2 accumulators (rows = output channels, columns = tokens) start from b2
        float4v acc2[TT][NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4v b2v = *(const float4v*)(p.b2 + j * 16 + g * 4);
#pragma unroll
            for (int i = 0; i < TT; ++i) acc2[i][j] = b2v;
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            // GEMM1 (transposed): acc1[ht][tt] = W1[32ch + 16ht ..][:] * Xn[16tt ..][:]^T   (rows = hidden, columns = tokens), from b1
            float4v acc1[2][TT];
            {
                const float4v be = *(const float4v*)(p.b1 + ch * 32 + g * 4);
                const float4v bo = *(const float4v*)(p.b1 + ch * 32 + 16 + g * 4);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) { acc1[0][tt] = be; acc1[1][tt] = bo; }
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int ks = i >> 1, ht = i & 1;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc1[ht][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[i], xreg[tt][ks], acc1[ht][tt], 0, 0, 0);
                wr[i] = *(const half8*)(WL + (size_t)(W1F + ch * NT + i) * 512);         // ring: W2 fragment i of this chunk, in flight under the GELU
                W2X_RING_FENCE();
            }
            // GELU in place; lane holds hidden rows 16ht + 4g + j of token column fr -> B fragment of GEMM2 for the k order
            // (ht 0: slots 0..3, ht 1: slots 4..7) that W2 is stored in
            half8 a2[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const float4v e = acc1[0][tt], o = acc1[1][tt];
                const float2v g0 = gelu_fast2((float2v){e[0], e[1]});
                const float2v g1 = gelu_fast2((float2v){e[2], e[3]});
                const float2v g2 = gelu_fast2((float2v){o[0], o[1]});
                const float2v g3 = gelu_fast2((float2v){o[2], o[3]});
                a2[tt] = (half8){(_Float16)g0[0], (_Float16)g0[1], (_Float16)g1[0], (_Float16)g1[1],
                                 (_Float16)g2[0], (_Float16)g2[1], (_Float16)g3[0], (_Float16)g3[1]};
            }
            // GEMM2 (transposed): acc2[tt][nt] += W2[16nt ..][chunk] * H[tokens][chunk]^T
            const int nch = ch + 1 < NCH ? ch + 1 : 0;          // after the last chunk: the first chunk's fragments for the next tile
#pragma unroll
            for (int i = 0; i < NF; ++i) {
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc2[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[i], a2[tt], acc2[tt][i], 0, 0, 0);
                wr[i] = *(const half8*)(WL + (size_t)((nch * 2 + (i & 1)) * KS + (i >> 1)) * 512);   // ring: W1 fragment i of the next chunk
                W2X_RING_FENCE();
            }
        }
        W2X_PHASE_FENCE();
        // ---- epilogue: residual pieces from the slab (raw rows), accumulators -> fp16 tile in the slab, then flat pieces
        half8 xres[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            xres[k] = *(const half8*)(Xw + r * LDX + c * 8);
        }
        W2X_PHASE_FENCE();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
                *(half4*)(Xw + (tt * 16 + fr) * LDX + nt * 16 + g * 4) = (half4){(_Float16)acc2[tt][nt][0], (_Float16)acc2[tt][nt][1], (_Float16)acc2[tt][nt][2], (_Float16)acc2[tt][nt][3]};
        W2X_PHASE_FENCE();
        _Float16* __restrict__ Y = (_Float16*)p.y + row0 * C;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            const half8 o = *(const half8*)(Xw + r * LDX + c * 8) + xres[k];     // fp16 + fp16 rounded once == fp32 add rounded to fp16
            if (idx < npieces) *(half8*)(Y + (size_t)idx * 8) = o;
            if (p.stats_out) *(half8*)(Xw + r * LDX + c * 8) = o;
        }
        W2X_PHASE_FENCE();
        if (p.stats_out && lane < nrows) {   // LayerNorm statistics of the produced rows for an un-fused consumer
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int c = 0; c < PPR; ++c) sum_sq8(*(const half8*)(Xw + lane * LDX + c * 8), s, q);
            const float mean = s * (1.f / C);
            p.stats_out[2 * (row0 + lane)] = mean;
            p.stats_out[2 * (row0 + lane) + 1] = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out);
        }
        W2X_PHASE_FENCE();   // the next tile's rows overwrite the slab
    }
}

}  // namespace

bool mlp96p_supported(const MlpParams& p) { return p.C == C && p.w1_frag && p.w2_frag; }

hipError_t launch_mlp96p(const MlpParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp96p_kernel, SMEM96P, lds_ok); e != hipSuccess) return e;
    static int cus[32] = {0};     // compute units per device (one resident workgroup each)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    int ncu = __atomic_load_n(&cus[dev & 31], __ATOMIC_RELAXED);
    if (ncu == 0) {
        if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        __atomic_store_n(&cus[dev & 31], ncu, __ATOMIC_RELAXED);
    }
    const long ntiles = (p.M + RW - 1) / RW;
    if (ntiles <= 0) return hipSuccess;
    if (ntiles > 0x7FFFFFFF) return hipErrorInvalidValue;
    const int grid = (int)std::min<long>((ntiles + NWV - 1) / NWV, ncu);
    hipLaunchKernelGGL(mlp96p_kernel, dim3(grid), dim3(NTHR), SMEM96P, s, p, (int)ntiles);
    return hipGetLastError();
}

}  // namespace w2x
