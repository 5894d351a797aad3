// Window multi-head self-attention on window-ordered rows (Swin W-MSA / SW-MSA core):
//   out[w][q][h*hd+d] = sum_k softmax_k( scale * q.k + bias[mask(w)][h][q][k] ) v[k][d]
// One wave per (window, head); lane q owns query row q (ntok = ws*ws <= 64 tokens): scores, softmax and the
// weighted sum all stay in that lane's registers, K/V rows are broadcast from LDS.  fp32 math on fp16 inputs.
// Covers the MatMul/Add/Softmax/MatMul chain inside TensorRT's enqueueV3 (img2img_infer.cpp:80).
#include "kernels.h"

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

}  // namespace

hipError_t launch_attn(const AttnParams& p, hipStream_t s) {
    long total = (long)p.B * p.nwin * p.heads;
    dim3 grid((unsigned)((total + 3) / 4));
    if (p.ntok == 36 && p.hd == 16) hipLaunchKernelGGL((attn_kernel<16, 36>), grid, dim3(256), 0, s, p);
    else if (p.ntok == 36 && p.hd == 32) hipLaunchKernelGGL((attn_kernel<32, 36>), grid, dim3(256), 0, s, p);
    else if (p.ntok == 64 && p.hd == 16) hipLaunchKernelGGL((attn_kernel<16, 64>), grid, dim3(256), 0, s, p);
    else if (p.ntok == 64 && p.hd == 32) hipLaunchKernelGGL((attn_kernel<32, 64>), grid, dim3(256), 0, s, p);
    else if (p.ntok == 36 && p.hd == 8) hipLaunchKernelGGL((attn_kernel<8, 36>), grid, dim3(256), 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace w2x
