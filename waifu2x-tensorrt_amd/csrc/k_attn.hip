// Window multi-head self-attention on window-ordered rows (Swin W-MSA / SW-MSA core) for graphs whose shapes the fused
// kernels (k_swinattn96.hip / k_swinattn192u.hip) do not cover.  Two kernels: attn_mfma_kernel (below, the default) on the matrix
// pipe, and the original lane-per-query VALU kernel (switches().attn_valu), kept as its reference.
//   out[w][q][h*hd+d] = sum_k softmax_k( scale * q.k + bias[mask(w)][h][q][k] ) v[k][d]
// One wave per (window, head); lane q owns query row q (ntok = ws*ws <= 64 tokens): scores, softmax and the
// weighted sum all stay in that lane's registers, K/V rows are broadcast from LDS.  fp32 math on fp16 inputs.
// Covers the MatMul/Add/Softmax/MatMul chain inside TensorRT's enqueueV3 (img2img_infer.cpp:80).
#include "kernels.h"

#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int HD, int NTOK>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) _Float16 sK[4][NTOK][HD];
    __shared__ __attribute__((aligned(16))) _Float16 sV[4][NTOK][HD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int C = p.heads * HD;
    const long total = (long)p.B * p.nwin * p.heads;
    long unit = (long)blockIdx.x * 4 + wv;
    const bool active_wave = unit < total;
    if (!active_wave) unit = total - 1;   // keep the wave alive for the barriers; results are not stored
    const int h = (int)(unit % p.heads);
    const long win = unit / p.heads;           // global window index (b*nwin + w)
    const int w = (int)(win % p.nwin);
    const bool act = lane < NTOK;
    const _Float16* base = (const _Float16*)p.qkv + (win * NTOK + (act ? lane : 0)) * (long)(3 * C) + h * HD;

    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
        half8 qv = *(const half8*)(base + d);
        half8 kv = *(const half8*)(base + C + d);
        half8 vv = *(const half8*)(base + 2 * C + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[d + e] = (float)qv[e] * p.scale;
        if (act) { *(half8*)&sK[wv][lane][d] = kv; *(half8*)&sV[wv][lane][d] = vv; }
    }
    __syncthreads();

    const _Float16* bias = (const _Float16*)p.bias + (((long)p.maskid[w] * p.heads + h) * NTOK + (act ? lane : 0)) * NTOK;
    float s[NTOK];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 8) {
            half8 kv = *(const half8*)&sK[wv][j][d];
#pragma unroll
            for (int e = 0; e < 8; ++e) a += q[d + e] * (float)kv[e];
        }
        a += (float)bias[j];
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
    const float inv = 1.f / sum;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        const float pj = s[j];
#pragma unroll
        for (int d = 0; d < HD; d += 8) {
            half8 vv = *(const half8*)&sV[wv][j][d];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[d + e] += pj * (float)vv[e];
        }
    }
    if (act && active_wave) {
        _Float16* op_ = (_Float16*)p.out + (win * NTOK + lane) * (long)C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 8) {
            half8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (_Float16)(o[d + e] * inv);
            *(half8*)(op_ + d) = ov;
        }
    }
}


// ---- the same core on the matrix pipe, for any head size in {8, 16, 32} and window of 36 or 64 tokens --------------------------
// One wave per (window, head) as above, but S^T = K Q^T and O^T = V^T P^T run on v_mfma_f32_16x16x16_f16: the accumulator
// layout of S^T (lane (query column, g) holds keys 4g..4g+3 of a key tile) is the B-operand layout of the second product, so P
// never leaves registers (the schedule of the fused kernels, k_swinattn96.hip, without their shape-specific tricks).  Operands
// come straight from the window-ordered qkv rows: a K / Q fragment is one 8-byte load per lane (4 consecutive features of one
// token), a V^T fragment four 2-byte loads (4 consecutive keys of one feature).  Rows beyond the window (36 tokens in three
// 16-row tiles) are loaded from the last token and masked: keys by a score of -inf, queries by not being stored.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int HD, int NTOK>
__global__ __launch_bounds__(256) void attn_mfma_kernel(const AttnParams p) {
    constexpr int NT = (NTOK + 15) / 16;          // token tiles
    constexpr int KS = (HD + 15) / 16;            // feature steps of 16 (HD = 8: half a step, upper features zero)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int C = p.heads * HD;
    const long total = (long)p.B * p.nwin * p.heads;
    const long unit = (long)blockIdx.x * 4 + wv;
    if (unit >= total) return;                    // no barrier in this kernel
    const int h = (int)(unit % p.heads);
    const long win = unit / p.heads;              // global window index (b*nwin + w)
    const int w = (int)(win % p.nwin);
    const _Float16* __restrict__ rows = (const _Float16*)p.qkv + win * NTOK * (long)(3 * C) + h * HD;
    const half4 zero4h = {};
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};

    // K (A operand: rows = keys) and Q (B operand: columns = queries) fragments: lane (token fr of the tile, g) holds features 16s + 4g .. +3
    half4 kf[NT][KS], qf[NT][KS];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tok = min(t * 16 + fr, NTOK - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool have = s * 16 + g * 4 < HD;
            const _Float16* src = rows + (long)tok * (3 * C) + s * 16 + (have ? g * 4 : 0);
            const half4 q = *(const half4*)src, k = *(const half4*)(src + C);
            qf[t][s] = have ? q : zero4h; kf[t][s] = have ? k : zero4h;
        }
    }
    // S^T tiles [key tile][query tile]
    float4v sc[NT][NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            float4v a = zero4;
#pragma unroll
            for (int s = 0; s < KS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt][s], qf[qt][s], a, 0, 0, 0);
            sc[kt][qt] = a;
        }
    // scale + bias (+ shift mask), in log2 units; keys beyond the window get -inf
    const float l2e = 1.44269504088896341f, qs = p.scale * l2e;
    const _Float16* __restrict__ bias = (const _Float16*)p.bias + ((long)p.maskid[w] * p.heads + h) * NTOK * NTOK;
    float mx[NT], sum[NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int query = min(qt * 16 + fr, NTOK - 1);
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int key0 = kt * 16 + g * 4;
            const half4 b = *(const half4*)(bias + (long)query * NTOK + (key0 < NTOK ? key0 : 0));     // NTOK is a multiple of 4: a group of 4 keys is inside the window or outside
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = key0 < NTOK ? fmaf(sc[kt][qt][j], qs, (float)b[j] * l2e) : -INFINITY;
                sc[kt][qt][j] = v;
                m = fmaxf(m, v);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));      // the four lanes that hold one query column
        mx[qt] = m;
    }
    // P^T = exp2(S^T - max) as fp16 B-operand fragments, row sums in fp32 over the fp16-rounded values
    half4 pf[NT][NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            half4 f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[j] = (_Float16)__builtin_amdgcn_exp2f(sc[kt][qt][j] - mx[qt]); l += (float)f[j]; }
            pf[kt][qt] = f;
        }
        l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
        sum[qt] = l;
    }
    // O^T = V^T P^T per 16-feature tile: A operand lane (feature fr, g) holds keys 4g..4g+3 of the key tile
    _Float16* __restrict__ out = (_Float16*)p.out + win * NTOK * (long)C + h * HD;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bool havef = s * 16 + fr < HD;
        float4v o[NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) o[qt] = zero4;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            half4 vf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = min(kt * 16 + g * 4 + j, NTOK - 1);            // keys beyond the window meet p = 0
                const _Float16 v = rows[(long)key * (3 * C) + 2 * C + s * 16 + (havef ? fr : 0)];
                vf[j] = havef ? v : (_Float16)0.f;
            }
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) o[qt] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf[kt][qt], o[qt], 0, 0, 0);
        }
        // lane (query column fr, g) holds features 16s + 4g .. +3 of its query
        if (s * 16 + g * 4 < HD) {
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                const int query = qt * 16 + fr;
                if (query < NTOK) {
                    const float inv = 1.f / sum[qt];
                    *(half4*)(out + (long)query * C + s * 16 + g * 4) = (half4){(_Float16)(o[qt][0] * inv), (_Float16)(o[qt][1] * inv), (_Float16)(o[qt][2] * inv), (_Float16)(o[qt][3] * inv)};
                }
            }
        }
    }
}

}  // namespace

hipError_t launch_attn(const AttnParams& p, hipStream_t s) {
    const bool valu = switches().attn_valu;   // reference path (switches.h): the lane-per-query VALU kernel above
    long total = (long)p.B * p.nwin * p.heads;
    dim3 grid((unsigned)((total + 3) / 4));
#define W2X_ATTN_CASE(HD_, NTOK_)                                                                                      \
    if (p.ntok == NTOK_ && p.hd == HD_) {                                                                               \
        if (valu) hipLaunchKernelGGL((attn_kernel<HD_, NTOK_>), grid, dim3(256), 0, s, p);                              \
        else hipLaunchKernelGGL((attn_mfma_kernel<HD_, NTOK_>), grid, dim3(256), 0, s, p);                              \
        return hipGetLastError();                                                                                       \
    }
    W2X_ATTN_CASE(16, 36) W2X_ATTN_CASE(32, 36) W2X_ATTN_CASE(16, 64) W2X_ATTN_CASE(32, 64) W2X_ATTN_CASE(8, 36)
#undef W2X_ATTN_CASE
    // other head sizes (multiples of 8 up to 64): the lane-per-query kernel, which is generic in both parameters
#define W2X_ATTN_VALU_CASE(HD_, NTOK_)                                                                                 \
    if (p.ntok == NTOK_ && p.hd == HD_) { hipLaunchKernelGGL((attn_kernel<HD_, NTOK_>), grid, dim3(256), 0, s, p); return hipGetLastError(); }
    W2X_ATTN_VALU_CASE(8, 64) W2X_ATTN_VALU_CASE(24, 36) W2X_ATTN_VALU_CASE(24, 64) W2X_ATTN_VALU_CASE(48, 36) W2X_ATTN_VALU_CASE(48, 64)
    W2X_ATTN_VALU_CASE(64, 36) W2X_ATTN_VALU_CASE(64, 64)
#undef W2X_ATTN_VALU_CASE
    return hipErrorInvalidValue;
}

// what launch_attn() takes: lower.cpp asks before it emits an OP_ATTN, so that a graph with another window or head size fails at build
// time naming the node instead of at the first render
}  // namespace w2x
