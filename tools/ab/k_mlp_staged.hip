// Fused Swin MLP branch for gfx950:   y = x + W2 * GELU(W1 * LayerNorm(x) + b1) + b2      (C = 96 or 192, hidden 2C)
// One launch replaces LayerNormalization + MatMul + Add + Div/Erf/Add/Mul/Mul + MatMul + Add + Add of the ONNX graph
// (the layers TensorRT runs inside enqueueV3, /root/reference/src/tensorrt/img2img_infer.cpp:80).
//
// Per workgroup (256 threads = 4 waves): 128 token rows.  x is read once from HBM, normalised into LDS (fp16; gamma/beta
// are folded into W1/b1), and the hidden activations never leave registers: GEMM1 is computed transposed
// (H^T = W1 * Xn^T) so that its 16x16 accumulator tiles - GELU applied in place - are exactly the A fragments of
// GEMM2 (k order permuted consistently on the W2 side: two 8-byte LDS reads per fragment).  W1/W2 stream through LDS in
// chunks of 64 hidden units with register prefetch of the next chunk.  The output tile is staged through LDS so the
// residual add and the HBM stores are 16-byte row pieces; LayerNorm statistics of the produced rows are emitted for
// the next op when requested.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// GELU(x) = 0.5 x (1 + erf(x/sqrt2)) = max(x,0) - 0.5 u erfc(u/sqrt2), u = |x|, with erfc(u/sqrt2) = 2^-q(u):
// q = -log2 erfc is smooth (~ u^2 log2(e)/2), so a degree-6 polynomial (tools/fit_gelu.py, weighted minimax fit on
// [0, 6.5]; beyond that the term is < 1e-9) gives |err| < 3.2e-7 absolute - the level of Abramowitz-Stegun 7.1.26 -
// with one quarter-rate transcendental (v_exp_f32) and packable fp32 FMAs instead of rcp + exp + sign fix-up.
__device__ __forceinline__ float gelu_fast(float x) {
    const float u = fminf(fabsf(x), 6.5f);
    float q = fmaf(-2.992485764e-05f, u, 7.398797018e-04f);
    q = fmaf(q, u, -7.977479093e-03f);
    q = fmaf(q, u, 5.323820859e-02f);
    q = fmaf(q, u, 4.589156733e-01f);
    q = fmaf(q, u, 1.151147085e+00f);
    const float e = __builtin_amdgcn_exp2f(-(q * u));
    return fmaf(-0.5f * u, e, fmaxf(x, 0.f));
}

// sum over aligned groups of LPR (16 or 32) lanes with DPP (no LDS crossbar): xor1, xor2, half-row mirror, row mirror
// packed-fp16 row helpers: sum and sum of squares of 8 halves through v_dot2_f32_f16 (fp32 accumulation), and the
// LayerNorm affine (x - mean) * rstd as one mixed-precision fma per element (fp16 in, fp32 scale/offset, one rounding)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sum_sq8(const half8 v, float& s, float& q) {
    const half2v one = {(_Float16)1.f, (_Float16)1.f};
    s = 0.f; q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const half2v h = {v[2 * k], v[2 * k + 1]};
        s = __builtin_amdgcn_fdot2(h, one, s, false);
        q = __builtin_amdgcn_fdot2(h, h, q, false);
    }
}
__device__ __forceinline__ half8 norm8(const half8 v, float rstd, float nm) {
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)fmaf((float)v[e], rstd, nm);
    return o;
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    if (LPR == 32) v += __shfl_xor(v, 16);
    return v;
}

// Diagnostic build only (W2X_STAMPS=1): per-phase s_memtime deltas summed over all waves.
// g_mlp_stamps[C==192][k]: 0 load+LN, 1 barriers+stage, 2 GEMM1, 3 GELU, 4 GEMM2, 5 epilogue, 7 waves
__device__ unsigned long long g_mlp_stamps[2][8];
#define W2X_MSTAMP(K)                                                                                       \
    if (STAMPS) {                                                                                           \
        unsigned long long t_;                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (lane == 0) atomicAdd(&g_mlp_stamps[C == 192][K], t_ - tprev);                                   \
        tprev = t_;                                                                                         \
    }

template <int C, bool STAMPS>
__global__ __launch_bounds__(256, (C == 96 ? 2 : 1)) void mlp_kernel(const MlpParams p) {
    constexpr int BM = 128, HC = 64;
    constexpr int LDX = C + 8;            // Xs row stride (halves)
    constexpr int LDW2 = HC + 8;          // W2s row stride
    constexpr int NCH = 2 * C / HC;       // hidden chunks
    constexpr int NT = C / 16;            // output n-tiles
    constexpr int LPR = C == 96 ? 16 : 32;  // lanes per row in the row-piece phases
    constexpr int PPR = C / 8;              // 16-byte pieces per row
    constexpr int W1_PIECES = HC * C / 8, W2_PIECES = C * HC / 8;
    constexpr int NW1 = W1_PIECES / 256, NW2 = W2_PIECES / 256;
    static_assert(W1_PIECES % 256 == 0 && W2_PIECES % 256 == 0, "piece counts");
    constexpr int LDC = C + 8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;                  // [BM][LDX]
    _Float16* W1s = Xs + BM * LDX;                   // [HC][LDX]
    _Float16* W2s = W1s + HC * LDX;                  // [C][LDW2]
    _Float16* Cs = W1s;                              // epilogue tile [BM][LDC] aliases the weight buffers (+ part of nothing else)
    static_assert((HC * LDX + C * LDW2) >= 0, "");

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long tprev = 0;
    if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); if (lane == 0) atomicAdd(&g_mlp_stamps[C == 192][7], 1ull); }
    const long row0 = (long)blockIdx.x * BM;
    const _Float16* __restrict__ X = (const _Float16*)p.x;
    const _Float16* __restrict__ W1 = (const _Float16*)p.w1;   // [2C][C]
    const _Float16* __restrict__ W2 = (const _Float16*)p.w2;   // [C][2C]

    // ---- weight chunk prefetch (global -> registers) and staging (registers -> LDS)
    u32x4 r1[NW1], r2[NW2];
#define W2X_MLP_PREFETCH(CH)                                                                                         \
    {                                                                                                                \
        _Pragma("unroll") for (int t = 0; t < NW1; ++t) {                                                            \
            const int idx = tid + t * 256, row = idx / PPR, kp = idx - row * PPR;                                    \
            r1[t] = *(const u32x4*)(W1 + (size_t)((CH) * HC + row) * C + kp * 8);                                    \
        }                                                                                                            \
        _Pragma("unroll") for (int t = 0; t < NW2; ++t) {                                                            \
            const int idx = tid + t * 256, row = idx / (HC / 8), kp = idx - row * (HC / 8);                          \
            r2[t] = *(const u32x4*)(W2 + (size_t)row * (2 * C) + (CH) * HC + kp * 8);                                \
        }                                                                                                            \
    }
#define W2X_MLP_STAGE()                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int t = 0; t < NW1; ++t) {                                                            \
            const int idx = tid + t * 256, row = idx / PPR, kp = idx - row * PPR;                                    \
            *(u32x4*)(W1s + row * LDX + kp * 8) = r1[t];                                                             \
        }                                                                                                            \
        _Pragma("unroll") for (int t = 0; t < NW2; ++t) {                                                            \
            const int idx = tid + t * 256, row = idx / (HC / 8), kp = idx - row * (HC / 8);                          \
            *(u32x4*)(W2s + row * LDW2 + kp * 8) = r2[t];                                                            \
        }                                                                                                            \
    }
    W2X_MLP_PREFETCH(0);

    // ---- LayerNorm of the 128 rows into Xs (LPR lanes per row, 8 channels per lane); all loads issued up front.
    //      (the raw rows are re-read for the residual in the epilogue: keeping them in 32 VGPRs measured 10% slower)
    constexpr int RPP = 256 / LPR, NPASS = BM / RPP;
    {
        const int li = tid & (LPR - 1);
        half8 xr[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const long row = row0 + ps * RPP + tid / LPR;
            half8 h = {};
            if (row < p.M && li < PPR) h = *(const half8*)(X + row * C + li * 8);
            xr[ps] = h;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            float s, q;
            sum_sq8(xr[ps], s, q);                       // lanes beyond the row hold zeros
            s = group_sum<LPR>(s);
            q = group_sum<LPR>(q);
            const float mean = s * (1.f / C);
            const float rstd = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps);
            if (li < PPR) *(half8*)(Xs + r * LDX + li * 8) = norm8(xr[ps], rstd, -mean * rstd);
        }
    }

    float4v acc2[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc2[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, g = lane >> 4;
    const int trow = wv * 32;   // this wave's 32 token rows

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {   // fully unrolled: prefetch registers are renamed statically, so the loads for chunk ch+1
                                         // are only waited for when that chunk is staged (a rolled loop makes hipcc wait right away)
        W2X_MSTAMP(ch == 0 ? 0 : 4)
        __syncthreads();
        W2X_MLP_STAGE();
        __syncthreads();
        W2X_MSTAMP(1)
        if (ch + 1 < NCH) W2X_MLP_PREFETCH(ch + 1);
        // GEMM1 (transposed): acc1[ht][tt] = W1s[16ht..][:] * Xs[trow+16tt..][:]^T   (rows = hidden, cols = tokens)
        float4v acc1[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[i][0] = (float4v){0.f, 0.f, 0.f, 0.f}; acc1[i][1] = (float4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
            half8 xb[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) xb[tt] = *(const half8*)(Xs + (trow + tt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                half8 wa = *(const half8*)(W1s + (ht * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc1[ht][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[tt], acc1[ht][tt], 0, 0, 0);
            }
        }
        W2X_MSTAMP(2)
        // bias + GELU in place; lane holds hidden rows 16ht + 4g + j of token column fr
        half8 a2[2][2];   // [tt][k-step]: A fragments of GEMM2, k order = (ht even: j 0..3, ht odd: j 4..7)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float4v be = *(const float4v*)(p.b1 + ch * HC + (2 * ks) * 16 + g * 4);
            const float4v bo = *(const float4v*)(p.b1 + ch * HC + (2 * ks + 1) * 16 + g * 4);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const float4v e = acc1[2 * ks][tt], o = acc1[2 * ks + 1][tt];
                a2[tt][ks] = (half8){(_Float16)gelu_fast(e[0] + be[0]), (_Float16)gelu_fast(e[1] + be[1]), (_Float16)gelu_fast(e[2] + be[2]),
                                     (_Float16)gelu_fast(e[3] + be[3]), (_Float16)gelu_fast(o[0] + bo[0]), (_Float16)gelu_fast(o[1] + bo[1]),
                                     (_Float16)gelu_fast(o[2] + bo[2]), (_Float16)gelu_fast(o[3] + bo[3])};
            }
        }
        W2X_MSTAMP(3)
        // GEMM2: acc2[tt][nt] += H[tokens][hidden chunk] * W2s[16nt..][chunk]^T with the matching k permutation
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const _Float16* wp = W2s + (nt * 16 + fr) * LDW2 + ks * 32 + g * 4;
                half4 lo = *(const half4*)wp, hi = *(const half4*)(wp + 16);
                half8 wb;
#pragma unroll
                for (int j = 0; j < 4; ++j) { wb[j] = lo[j]; wb[4 + j] = hi[j]; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc2[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[tt][ks], wb, acc2[tt][nt], 0, 0, 0);
            }
        }
    }
    W2X_MSTAMP(4)
    __syncthreads();

    // ---- epilogue: residual pieces are fetched first (latency overlaps the tile write), accumulators + b2 -> fp16
    //      tile in LDS, then row pieces: + residual x, store, statistics
    const int li = tid & (LPR - 1);
    half8 xres[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const long row = row0 + ps * RPP + tid / LPR;
        half8 h = {};
        if (row < p.M && li < PPR) h = *(const half8*)(X + row * C + li * 8);
        xres[ps] = h;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b2 = p.b2[nt * 16 + fr];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) Cs[(trow + tt * 16 + g * 4 + j) * LDC + nt * 16 + fr] = (_Float16)(acc2[tt][nt][j] + b2);
    }
    __syncthreads();
    {
        _Float16* __restrict__ Y = (_Float16*)p.y;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            const long row = row0 + r;
            const bool ok = row < p.M && li < PPR;
            half8 o = {};
            if (ok) {
                o = *(const half8*)(Cs + r * LDC + li * 8) + xres[ps];     // fp16 + fp16 rounded once == fp32 add rounded to fp16
                *(half8*)(Y + row * C + li * 8) = o;
            }
            if (p.stats_out) {
                float s, q;
                sum_sq8(o, s, q);
                s = group_sum<LPR>(s);
                q = group_sum<LPR>(q);
                const float mean = s * (1.f / C);
                if (ok && li == 0) { p.stats_out[2 * row] = mean; p.stats_out[2 * row + 1] = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out); }
            }
        }
    }
    W2X_MSTAMP(5)
}

template <int C>
hipError_t launch_mlp_c(const MlpParams& p, hipStream_t s) {
    constexpr int BM = 128, HC = 64, LDX = C + 8, LDW2 = HC + 8, LDC = C + 8;
    constexpr int W_BYTES = (HC * LDX + C * LDW2) * 2, C_BYTES = BM * LDC * 2;
    constexpr int SMEM = BM * LDX * 2 + (W_BYTES > C_BYTES ? W_BYTES : C_BYTES);
    static const bool stamps = getenv("W2X_STAMPS") != nullptr;
    auto kern = stamps ? mlp_kernel<C, true> : mlp_kernel<C, false>;
    static unsigned lds_ok = 0, lds_ok2 = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp_kernel<C, true>, SMEM, lds_ok); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp_kernel<C, false>, SMEM, lds_ok2); e != hipSuccess) return e;
    dim3 grid((unsigned)((p.M + BM - 1) / BM));
    hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t read_mlp_stamps(unsigned long long* out) {   // diagnostic: 16 values, cleared on read
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_stamps), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return e;
    unsigned long long z[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_stamps), z, sizeof(z));
}

hipError_t launch_mlp2(const MlpParams& p, hipStream_t s);   // k_mlp2.hip

hipError_t launch_mlp(const MlpParams& p, hipStream_t s) {
    static const bool v1 = getenv("W2X_MLP_V1") != nullptr;   // A/B switch: the LDS-staged kernel of this file
    if (!v1 && p.w1_frag && p.w2_frag && (p.C == 96 || p.C == 192)) return launch_mlp2(p, s);
    if (p.C == 96) return launch_mlp_c<96>(p, s);
    if (p.C == 192) return launch_mlp_c<192>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace w2x
