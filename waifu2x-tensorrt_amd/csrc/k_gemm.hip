// Fused implicit-GEMM kernel for gfx950 (CDNA4): every Conv / ConvTranspose / MatMul of the cunet and swin_unet
// graphs runs through this one template.  It replaces the convolution / matmul layers TensorRT would pick for
// IExecutionContext::enqueueV3 (/root/reference/src/tensorrt/img2img_infer.cpp:80).
//
//   Y[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] )
//   A rows   : pixels of a channel-last fp16 map - plain, gathered through a window table (Swin window
//              partition + cyclic shift), or kh x kw convolution taps (im2col-free: addresses only)
//   epilogue : folded LayerNorm (rstd*(acc - mean*csum)), bias, LeakyReLU/GELU/ReLU/sigmoid, up to two residual
//              adds, clip, pixel-shuffle / window scatter stores, LayerNorm statistics of the produced rows for
//              the next op, per-channel sums for squeeze-excite.
// Tiling: 256 threads = 4 waves (64 lanes), v_mfma_f32_16x16x32_f16, A/B sub-chunks of KB columns staged through
// LDS with register prefetch of the next sub-chunk, results staged through LDS so HBM stores are 16-byte rows.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_fn(float v, int act, float alpha) {
    switch (act) {
        case 1: return v > 0.f ? v : v * alpha;
        case 2: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case 3: return v > 0.f ? v : 0.f;
        case 4: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

template <int KB, int WAVES_M, int WAVES_N, int WM, int WN, int AP, int OP>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    constexpr int BM = WAVES_M * WM * 16;
    constexpr int BN = WAVES_N * WN * 16;
    constexpr int LDA = KB + 8;                 // halves; +16 B pad keeps ds_read_b128 (mostly) conflict free
    constexpr int LDC = BN + 8;
    constexpr int A_PIECES = BM * KB / AP;      // pieces of AP halves per sub-chunk
    constexpr int B_PIECES = BN * KB / 8;
    constexpr int NA = (A_PIECES + 255) / 256;
    constexpr int NB = (B_PIECES + 255) / 256;
    constexpr int A_PER_ROW = KB / AP, B_PER_ROW = KB / 8;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(KB % 32 == 0, "KB multiple of the MFMA K");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* As = (_Float16*)smem;
    _Float16* Bs = As + BM * LDA;
    _Float16* Cs = (_Float16*)smem;  // aliases As/Bs after the main loop
    constexpr int AB_BYTES = (BM + BN) * LDA * 2;
    constexpr int C_BYTES = BM * LDC * 2;
    constexpr int MAIN_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    int* s_aoff = (int*)(smem + MAIN_BYTES);
    int* s_ob = s_aoff + BM;
    int* s_oy = s_ob + BM;
    int* s_ox = s_oy + BM;
    float* s_mean = (float*)(s_ox + BM);
    float* s_rstd = s_mean + BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;
    // row tiles never straddle batch items: blockIdx.x = b * tiles_per_image + tile (keeps every per-image reduction
    // - squeeze-excite pooling - independent of the batch slot a tile sits in)
    const int tpi = (p.Mrows + BM - 1) / BM;
    const int tile_b = blockIdx.x / tpi;
    const int ml0 = (blockIdx.x - tile_b * tpi) * BM;
    const int n0 = blockIdx.y * BN;
    const _Float16* __restrict__ Ag = (const _Float16*)p.a.p;
    const _Float16* __restrict__ Wg = (const _Float16*)p.wt;

    // ---- per-row bookkeeping
    for (int i = tid; i < BM; i += 256) {
        int ml = ml0 + i;
        int aoff = -1, ob = 0, oy = 0, ox = 0;
        float mean = 0.f, rstd = 1.f;
        if (ml < p.Mrows) {
            int b = tile_b;
            int y, x;
            if (p.amode == 1) { int pix = p.win_table[ml]; y = pix / p.aW; x = pix - y * p.aW; }
            else { y = ml / p.aW; x = ml - y * p.aW; }
            int pixoff = (b * p.a.Hs + y * p.stride + p.a.y0) * p.a.Ws + x * p.stride + p.a.x0;
            aoff = pixoff * p.a.Cs;
            if (p.ln) { mean = p.stats_in[2 * pixoff]; rstd = p.stats_in[2 * pixoff + 1]; }
            ob = b;
            if (p.omode == 1) { int pix = p.win_table[ml]; oy = pix / p.out.Ws; ox = pix - oy * p.out.Ws; }
            else { oy = ml / p.aW; ox = ml - oy * p.aW; }
        }
        s_aoff[i] = aoff; s_ob[i] = ob; s_oy[i] = oy; s_ox[i] = ox; s_mean[i] = mean; s_rstd[i] = rstd;
    }
    __syncthreads();

    // ---- register staging of one sub-chunk
    uint4 ra[NA];  // AP == 4 uses only .x/.y
    uint4 rb[NB];
    const int Cin = p.a.Cs;
    const int kwCin = p.kw * Cin;
    auto load_regs = [&](int kc) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            int idx = tid + t * 256;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (A_PIECES % 256 == 0 || idx < A_PIECES) {
                int row = idx / A_PER_ROW, kp = idx - row * A_PER_ROW;
                int k = kc + kp * AP;
                int aoff = s_aoff[row];
                if (aoff >= 0 && k < p.K) {
                    int off;
                    if (p.amode == 2) { int ky = k / kwCin; int rem = k - ky * kwCin; off = aoff + ky * p.a.Ws * Cin + rem; }
                    else off = aoff + k;
                    if (AP == 8) {
                        v = *(const uint4*)(Ag + off);
                        if (p.a_scale) {   // squeeze-excite gate of the input map: fp16(x * s), the rounding of the in-place pass
                            const float* sc = p.a_scale + tile_b * Cin + (p.amode == 2 ? (k % kwCin) % Cin : k);
                            v = __builtin_bit_cast(uint4, gate::gate8(__builtin_bit_cast(half8, v), sc));
                        }
                    } else { uint2 u = *(const uint2*)(Ag + off); v.x = u.x; v.y = u.y; }
                }
            }
            ra[t] = v;
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            int idx = tid + t * 256;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (B_PIECES % 256 == 0 || idx < B_PIECES) {
                int row = idx / B_PER_ROW, kp = idx - row * B_PER_ROW;
                int k = kc + kp * 8, n = n0 + row;
                if (n < p.N && k < p.Kw) v = *(const uint4*)(Wg + (size_t)n * p.Kw + k);
            }
            rb[t] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            int idx = tid + t * 256;
            if (A_PIECES % 256 == 0 || idx < A_PIECES) {
                int row = idx / A_PER_ROW, kp = idx - row * A_PER_ROW;
                if (AP == 8) *(uint4*)(As + row * LDA + kp * 8) = ra[t];
                else *(uint2*)(As + row * LDA + kp * 4) = make_uint2(ra[t].x, ra[t].y);
            }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            int idx = tid + t * 256;
            if (B_PIECES % 256 == 0 || idx < B_PIECES) {
                int row = idx / B_PER_ROW, kp = idx - row * B_PER_ROW;
                *(uint4*)(Bs + row * LDA + kp * 8) = rb[t];
            }
        }
    };

    float4v acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int nchunks = (p.K + KB - 1) / KB;
    const int frow = lane & 15, fk = (lane >> 4) * 8;
    load_regs(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_lds();
        __syncthreads();
        if (c + 1 < nchunks) load_regs((c + 1) * KB);
#pragma unroll
        for (int ks = 0; ks < KB / 32; ++ks) {
            half8 af[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const half8*)(As + ((wm * WM + i) * 16 + frow) * LDA + ks * 32 + fk);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                half8 bf = *(const half8*)(Bs + ((wn * WN + j) * 16 + frow) * LDA + ks * 32 + fk);
#pragma unroll
                for (int i = 0; i < WM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- epilogue phase 1: accumulators -> (LayerNorm algebra, bias, activation) -> fp16 tile in LDS
    {
        const int ccol = lane & 15, crow = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            int col = (wn * WN + j) * 16 + ccol;
            int n = n0 + col;
            float bias = 0.f, cs = 0.f;
            if (n < p.N) { bias = p.bias ? p.bias[n] : 0.f; if (p.ln) cs = p.csum[n]; }
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int row = (wm * WM + i) * 16 + crow + e;
                    float v = acc[i][j][e];
                    if (p.ln) v = s_rstd[row] * (v - s_mean[row] * cs);
                    v += bias;
                    v = act_fn(v, p.act, p.alpha);
                    Cs[row * LDC + col] = (_Float16)v;
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue phase 2: rows of OP halves: residual adds, clip, store, LayerNorm statistics, SE pooling
    {
        _Float16* __restrict__ Og = (_Float16*)p.out.p;
        const int Cso = p.out.Cs;
        const bool pix = p.omode == 2;
        const int ppc = (pix ? Cso : BN) / OP;                 // pieces per (row, sub-pixel) in this chunk
        const int subs = pix ? BN / Cso : 1;
        int gs = 1; while (gs < ppc) gs <<= 1;
        const int groups = 256 / gs;
        const int items = BM * subs;
        const int jp = tid & (gs - 1);
        for (int q0 = 0; q0 < items; q0 += groups) {
            int q = q0 + tid / gs;
            if (q >= items) q = items - 1;   // clamp (duplicates are masked by `valid` below)
            int i = q / subs, s = q - i * subs;
            bool valid = (q0 + tid / gs) < items && jp < ppc && s_aoff[i] >= 0;
            int ccol = pix ? s * Cso + jp * OP : jp * OP;        // column inside the LDS tile
            int ch = pix ? jp * OP : n0 + jp * OP;                // channel inside the output pixel
            int sg = pix ? n0 / Cso + s : 0;
            if (pix ? sg >= p.r * p.r : ch >= p.N) valid = false;
            int dy = pix ? sg / p.r : 0, dx = pix ? sg - dy * p.r : 0;
            float v[OP];
#pragma unroll
            for (int e = 0; e < OP; ++e) v[e] = 0.f;
            int b = 0, Y = 0, X = 0;
            if (valid) {
                b = s_ob[i]; Y = s_oy[i] * p.r + dy; X = s_ox[i] * p.r + dx;
#pragma unroll
                for (int e = 0; e < OP; ++e) v[e] = (float)Cs[i * LDC + ccol + e];
                if (p.res.p) {
                    const _Float16* rp = (const _Float16*)p.res.p + (size_t)((b * p.res.Hs + Y + p.res.y0) * p.res.Ws + X + p.res.x0) * p.res.Cs + ch;
                    if (OP == 8) { half8 r = *(const half8*)rp;
                        if (p.res_scale) {
                            const float* rs = p.res_scale + b * p.res.Cs + ch;
                            r = gate::gate8(r, rs);
                        }
#pragma unroll
                        for (int e = 0; e < OP; ++e) v[e] += (float)r[e]; }
                    else { half4 r = *(const half4*)rp;
#pragma unroll
                        for (int e = 0; e < OP; ++e) v[e] += (float)r[e]; }
                }
                if (p.res2.p) {
                    const _Float16* rp = (const _Float16*)p.res2.p + (size_t)((b * p.res2.Hs + Y + p.res2.y0) * p.res2.Ws + X + p.res2.x0) * p.res2.Cs + ch;
                    if (OP == 8) { half8 r = *(const half8*)rp;
#pragma unroll
                        for (int e = 0; e < OP; ++e) v[e] += (float)r[e]; }
                    else { half4 r = *(const half4*)rp;
#pragma unroll
                        for (int e = 0; e < OP; ++e) v[e] += (float)r[e]; }
                }
                if (p.has_clip) {
#pragma unroll
                    for (int e = 0; e < OP; ++e) v[e] = fminf(fmaxf(v[e], p.clip_lo), p.clip_hi);
                }
                _Float16 h[OP];
#pragma unroll
                for (int e = 0; e < OP; ++e) { h[e] = (_Float16)v[e]; v[e] = (float)h[e]; }
                _Float16* op_ = Og + (size_t)((b * p.out.Hs + Y) * p.out.Ws + X) * Cso + ch;
                if (OP == 8) *(uint4*)op_ = *(const uint4*)h; else *(uint2*)op_ = *(const uint2*)h;
                if (p.pool_out) {   // keep the final (fp16-rounded) values for the deterministic column sums below
                    if (OP == 8) *(uint4*)(Cs + i * LDC + ccol) = *(const uint4*)h; else *(uint2*)(Cs + i * LDC + ccol) = *(const uint2*)h;
                }
            }
            if (p.stats_out) {   // uniform branch; all lanes take part in the shuffles
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < OP; ++e) sum += v[e];
                for (int msk = gs >> 1; msk > 0; msk >>= 1) sum += __shfl_xor(sum, msk);
                float mean = sum / (float)p.Cout;
                float sq = 0.f;
                if (valid) {
#pragma unroll
                    for (int e = 0; e < OP; ++e) { float d = v[e] - mean; sq += d * d; }
                }
                for (int msk = gs >> 1; msk > 0; msk >>= 1) sq += __shfl_xor(sq, msk);
                if (valid && jp == 0) {
                    size_t pixi = (size_t)(b * p.out.Hs + Y) * p.out.Ws + X;
                    p.stats_out[2 * pixi] = mean;
                    p.stats_out[2 * pixi + 1] = rsqrtf(sq / (float)p.Cout + p.ln_eps);
                }
            }
        }
        if (p.pool_out) {
            // squeeze-excite pooling without atomics: per-workgroup column sums in row order;
            // se_kernel adds the partials of a batch item's workgroups in tile order.
            __syncthreads();
            for (int c = tid; c < BN; c += 256) {
                float sum = 0.f;
                for (int i = 0; i < BM; ++i) if (s_aoff[i] >= 0) sum += (float)Cs[i * LDC + c];
                if (n0 + c < p.N) p.pool_out[(size_t)blockIdx.x * Cso + n0 + c] = sum;
            }
        }
    }
}

template <int KB, int WAVES_M, int WAVES_N, int WM, int WN, int AP, int OP>
hipError_t launch_cfg(const GemmParams& p, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16, LDA = KB + 8, LDC = BN + 8;
    constexpr int AB = (BM + BN) * LDA * 2, CB = BM * LDC * 2;
    constexpr int MAIN = AB > CB ? AB : CB;
    constexpr int SMEM = MAIN + BM * 6 * 4;
    static_assert(BM == kGemmBM, "plan.h sizes the pooling partials for this tile height");
    auto kern = gemm_kernel<KB, WAVES_M, WAVES_N, WM, WN, AP, OP>;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, SMEM, lds_ok); e != hipSuccess) return e;
    dim3 grid(p.B * ((p.Mrows + BM - 1) / BM), (p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

// Tile selection.  BN follows N (all widths in these networks are multiples of 16 after padding); KB = 96 when it
// divides K (Swin C = 96/192, 3x3 convs over 32/64/128/256 channels), else 64.
hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    if ((p.a_scale && (p.a.Cs == 4 || p.a.Cs % 8)) || (p.res_scale && (p.out.Cs == 4 || !p.res.p))) return hipErrorInvalidValue;   // gates ride on 8-half pieces
    const bool op4 = p.out.Cs == 4;
    if (p.a.Cs == 4) {                       // first convolution: 4 stored input channels, K = 36
        if (p.N <= 32) return launch_cfg<64, 4, 1, 2, 2, 4, 8>(p, s);
        if (p.N <= 48) return launch_cfg<64, 4, 1, 2, 3, 4, 8>(p, s);
        if (p.N <= 64) return launch_cfg<64, 4, 1, 2, 4, 4, 8>(p, s);
        return hipErrorInvalidValue;
    }
    if (op4) {                               // 3-channel heads (4 stored channels per pixel)
        if (p.N <= 16) return (p.K % 96 == 0) ? launch_cfg<96, 4, 1, 2, 1, 8, 4>(p, s) : launch_cfg<64, 4, 1, 2, 1, 8, 4>(p, s);
        if (p.N <= 64) return (p.K % 96 == 0) ? launch_cfg<96, 4, 1, 2, 4, 8, 4>(p, s) : launch_cfg<64, 4, 1, 2, 4, 8, 4>(p, s);
        return hipErrorInvalidValue;
    }
    const bool k96 = p.K % 96 == 0;
    int bn;
    if (p.omode == 2) bn = (p.out.Cs <= 192 && 192 % p.out.Cs == 0 && p.N >= 192) ? 192 : p.out.Cs;
    else bn = p.N % 192 == 0 ? 192 : p.N % 96 == 0 ? 96 : p.N % 128 == 0 ? 128 : p.N % 64 == 0 ? 64 : p.N % 48 == 0 ? 48 : p.N % 32 == 0 ? 32 : 0;
    // pixel-shuffle projections onto 96-channel pixels are memory-bound: 96-wide tiles (150 VGPRs, 3 workgroups per CU) keep
    // more rows in flight than the 192-wide ones (244 VGPRs, 2 per CU): 0.69 -> 0.61 ms on config 3's last up-projection
    if (bn == 192 && !p.stats_out && p.omode == 2 && 96 % p.out.Cs == 0) bn = 96;
    if (p.stats_out && p.omode != 2 && bn < p.N) return hipErrorInvalidValue;
    switch (bn) {
        case 192: return k96 ? launch_cfg<96, 2, 2, 4, 6, 8, 8>(p, s) : launch_cfg<64, 2, 2, 4, 6, 8, 8>(p, s);
        case 128: return k96 ? launch_cfg<96, 2, 2, 4, 4, 8, 8>(p, s) : launch_cfg<64, 2, 2, 4, 4, 8, 8>(p, s);
        case 96: return k96 ? launch_cfg<96, 4, 1, 2, 6, 8, 8>(p, s) : launch_cfg<64, 4, 1, 2, 6, 8, 8>(p, s);
        case 64: return k96 ? launch_cfg<96, 4, 1, 2, 4, 8, 8>(p, s) : launch_cfg<64, 4, 1, 2, 4, 8, 8>(p, s);
        case 48: return k96 ? launch_cfg<96, 4, 1, 2, 3, 8, 8>(p, s) : launch_cfg<64, 4, 1, 2, 3, 8, 8>(p, s);
        case 32: return k96 ? launch_cfg<96, 4, 1, 2, 2, 8, 8>(p, s) : launch_cfg<64, 4, 1, 2, 2, 8, 8>(p, s);
        default: return hipErrorInvalidValue;
    }
}

// widths whose rows fit one workgroup tile of launch_gemm() (bn >= N): only those can carry LayerNorm statistics out of the epilogue
}  // namespace w2x
