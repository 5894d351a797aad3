// The engine's second storage type for gfx950: every Conv / ConvTranspose / MatMul and the window attention core on fp32 maps.
// The reference offers fp16 and TF32 engines (/root/reference/src/tensorrt/config.h:7-10, img2img_build.cpp:123-135); gfx950 has
// no TF32 / xf32 matrix instruction.  An fp32 plan (plan.h Plan::elt == 4, lower.cpp) keeps fp32 maps, weights and accumulation and
// the un-fused operator set (no fused transformer kernels, no shape-specialised convolutions), and serves two precisions
// (include/w2x/config.h):
//   Precision::TF32  TF32-CLASS products.  Both operands are split when they are staged into LDS, x = hi + lo with hi = bf16(x),
//                    lo = bf16(x - hi) (16 significant bits, fp32's exponent range), and a k-step of 32 is three
//                    v_mfma_f32_16x16x32_bf16: lo*hi + hi*lo + hi*hi (the dropped lo*lo is 2^-18 of the product).  Against the fp32 oracle
//                    8.9e-6 at most (mean 1.4e-6) on outputs in [0, 1]; eleven-bit operands - what TF32 keeps - would give about 1e-3.
//   Precision::FP32  exact fp32 products on v_mfma_f32_16x16x4_f32 (an addition to the reference's enum): 6.6e-7 at most, summation order.
// Two kernels cover the plan:
//   gemm32_kernel : gemm_kernel (k_gemm.hip) restated for fp32 maps - same implicit-GEMM A operand (rows / window gather / kh x kw
//                   taps), same epilogue (folded LayerNorm, bias, activations, two residual adds, clip, rows / window / pixel-shuffle
//                   stores, LayerNorm statistics and squeeze-excite partial sums of the stored rows), same 128-row workgroup tile
//                   (kGemmBM), LDS-staged operands with register prefetch of the next k-chunk.
//   attn32_kernel : one wave per (window, head), lane = query row, everything in that lane's registers (k_attn.hip attn_kernel
//                   with fp32 rows).
// Config 3 (1080p frame, round 5): 49.8 ms with TF32, 64.7 ms with FP32 (7.4 ms on the fp16 engine).  Until round 5 both were one 70 ms
// path; what moved it: the split products (-10 ms) and an epilogue whose piece geometry is a compile-time fact when the launch is
// not a pixel shuffle (its index arithmetic was a third of the kernel's vector instructions: -5 ms / -8 ms), and with TF32 GELU as a
// polynomial around v_exp_f32 instead of erff() (-2 ms).  What is left is the
// fp32 traffic of an un-fused plan (2 - 3 TB/s per launch) and the lane-per-query attention core (10.8 ms).
#include "kernels.h"
#include <cmath>
#include <type_traits>

namespace w2x {
namespace {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = hi + lo + e with hi = bf16(x), lo = bf16(x - hi) (the difference is exact in fp32), |e| <= 2^-18 |x|: four values -> 4 + 4 bf16
__device__ __forceinline__ void split4(const float4v v, uint2v& hi, uint2v& lo) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float a = v[2 * h], b = v[2 * h + 1];
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector((float2v){a, b}, bf16x2));   // v_cvt_pk_bf16_f32, round to nearest even
        const float ah = __builtin_bit_cast(float, hb << 16), bh = __builtin_bit_cast(float, hb & 0xFFFF0000u);
        hi[h] = hb;
        lo[h] = __builtin_bit_cast(unsigned, __builtin_convertvector((float2v){a - ah, b - bh}, bf16x2));
    }
}

__device__ __forceinline__ float act_fn(float v, int act, float alpha) {
    switch (act) {
        case 1: return v > 0.f ? v : v * alpha;
        case 2: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case 3: return v > 0.f ? v : 0.f;
        case 4: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// GELU for the split-product (TF32) path: max(x, 0) - 0.5 u 2^-q(u), u = min(|x|, 6.5), six coefficients (tools/fit_gelu.py: |err| < 3.1e-7, a thirtieth of what the
// split products leave) - 10 instructions where erff() takes about 35; a full-resolution fc1 launch of config 3 is 0.5 G values.
__device__ __forceinline__ float gelu_poly(float x) {
    const float u = fminf(fabsf(x), 6.5f);
    float q = fmaf(-2.992485764e-05f, u, 7.398797018e-04f);
    q = fmaf(q, u, -7.977479093e-03f);
    q = fmaf(q, u, 5.323820859e-02f);
    q = fmaf(q, u, 4.589156733e-01f);
    q = fmaf(q, u, 1.151147085e+00f);
    return fmaf(-0.5f * u, __builtin_amdgcn_exp2f(-(q * u)), fmaxf(x, 0.f));
}

#ifndef W2X_G32_WPC
#define W2X_G32_WPC 3      // workgroups per CU the register budget is set for (the half-height epilogue tile leaves LDS for three)
#endif
// SPLIT (the default, see the head of the file): operands staged as bf16 hi / lo planes, three v_mfma_f32_16x16x32_bf16 per k-step of 32;
// !SPLIT: fp32 operands, v_mfma_f32_16x16x4_f32.  Gather, prefetch and epilogue are the same code.
// PIX: pixel-shuffle stores (omode 2).  Without them the piece geometry of the epilogue is a compile-time fact (no sub-pixels, BN / 4 pieces per row) and its index
// arithmetic folds to shifts and masks - it used to be a third of the kernel's vector instructions.
template <int WAVES_M, int WAVES_N, int WM, int WN, int HALVES, bool SPLIT, bool PIX>
__global__ __launch_bounds__(256, SPLIT && WM * WN > 16 ? 2 : W2X_G32_WPC) void gemm32_kernel(const GemmParams p) {   // (the 128 x 192 tile with split operands: 96 accumulators + 40 of prefetch + 40 of fragments)
    constexpr int KB = SPLIT ? 32 : 16;         // k-chunk: one bf16 k-step / four v_mfma_f32_16x16x4_f32 steps
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16;
    constexpr int LDA = KB + 1, LDC = BN + 4;   // floats
    constexpr int LDH = KB + 8;                 // bf16 per row of a split plane: 80 bytes, 16-byte aligned, conflict-free ds_read_b128
    constexpr int A_PIECES = BM * KB / 4, B_PIECES = BN * KB / 4;   // pieces of 4 floats
    constexpr int NA = (A_PIECES + 255) / 256, NB = (B_PIECES + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4 && BM == kGemmBM, "4 waves, kGemmBM rows");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* As = (float*)smem;
    float* Bs = As + BM * LDA;
    float* Cs = (float*)smem;                   // aliases As / Bs after the main loop
    unsigned short* AsH = (unsigned short*)smem;                        // SPLIT: [BM][LDH] hi, [BM][LDH] lo, [BN][LDH] hi, [BN][LDH] lo
    unsigned short* AsL = AsH + BM * LDH;
    unsigned short* BsH = AsL + BM * LDH;
    unsigned short* BsL = BsH + BN * LDH;
    constexpr int AB_BYTES = SPLIT ? (BM + BN) * LDH * 4 : (BM + BN) * LDA * 4, C_BYTES = BM / HALVES * LDC * 4;
    constexpr int MAIN_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    int* s_aoff = (int*)(smem + MAIN_BYTES);
    int* s_ob = s_aoff + BM;
    int* s_oy = s_ob + BM;
    int* s_ox = s_oy + BM;
    float* s_mean = (float*)(s_ox + BM);
    float* s_rstd = s_mean + BM;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;
    const int tpi = (p.Mrows + BM - 1) / BM;    // row tiles never straddle batch items
    const int tile_b = blockIdx.x / tpi;
    const int ml0 = (blockIdx.x - tile_b * tpi) * BM;
    const int n0 = blockIdx.y * BN;
    const float* __restrict__ Ag = (const float*)p.a.p;
    const float* __restrict__ Wg = (const float*)p.wt;

    for (int i = tid; i < BM; i += 256) {
        const int ml = ml0 + i;
        int aoff = -1, ob = 0, oy = 0, ox = 0;
        float mean = 0.f, rstd = 1.f;
        if (ml < p.Mrows) {
            const int b = tile_b;
            int y, x;
            if (p.amode == 1) { const int pix = p.win_table[ml]; y = pix / p.aW; x = pix - y * p.aW; }
            else { y = ml / p.aW; x = ml - y * p.aW; }
            const int pixoff = (b * p.a.Hs + y * p.stride + p.a.y0) * p.a.Ws + x * p.stride + p.a.x0;
            aoff = pixoff * p.a.Cs;
            if (p.ln) { mean = p.stats_in[2 * pixoff]; rstd = p.stats_in[2 * pixoff + 1]; }
            ob = b;
            if (p.omode == 1) { const int pix = p.win_table[ml]; oy = pix / p.out.Ws; ox = pix - oy * p.out.Ws; }
            else { oy = ml / p.aW; ox = ml - oy * p.aW; }
        }
        s_aoff[i] = aoff; s_ob[i] = ob; s_oy[i] = oy; s_ox[i] = ox; s_mean[i] = mean; s_rstd[i] = rstd;
    }
    __syncthreads();

    float4v ra[NA], rb[NB];
    const int Cin = p.a.Cs, kwCin = p.kw * Cin;
    auto load_regs = [&](int kc) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int idx = tid + t * 256;
            float4v v = {0.f, 0.f, 0.f, 0.f};
            if (A_PIECES % 256 == 0 || idx < A_PIECES) {
                const int row = idx / (KB / 4), kp = idx - row * (KB / 4), k = kc + kp * 4, aoff = s_aoff[row];
                if (aoff >= 0 && k < p.K) {
                    int off;
                    if (p.amode == 2) { const int ky = k / kwCin, rem = k - ky * kwCin; off = aoff + ky * p.a.Ws * Cin + rem; }
                    else off = aoff + k;
                    v = *(const float4v*)(Ag + off);
                }
            }
            ra[t] = v;
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int idx = tid + t * 256;
            float4v v = {0.f, 0.f, 0.f, 0.f};
            if (B_PIECES % 256 == 0 || idx < B_PIECES) {
                const int row = idx / (KB / 4), kp = idx - row * (KB / 4), k = kc + kp * 4, n = n0 + row;
                if (n < p.N && k < p.Kw) v = *(const float4v*)(Wg + (size_t)n * p.Kw + k);
            }
            rb[t] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int idx = tid + t * 256;
            if (A_PIECES % 256 == 0 || idx < A_PIECES) {
                const int row = idx / (KB / 4), kp = idx - row * (KB / 4);
                if (SPLIT) {
                    uint2v hi, lo;
                    split4(ra[t], hi, lo);
                    *(uint2v*)(AsH + row * LDH + kp * 4) = hi;
                    *(uint2v*)(AsL + row * LDH + kp * 4) = lo;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) As[row * LDA + kp * 4 + e] = ra[t][e];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int idx = tid + t * 256;
            if (B_PIECES % 256 == 0 || idx < B_PIECES) {
                const int row = idx / (KB / 4), kp = idx - row * (KB / 4);
                if (SPLIT) {
                    uint2v hi, lo;
                    split4(rb[t], hi, lo);
                    *(uint2v*)(BsH + row * LDH + kp * 4) = hi;
                    *(uint2v*)(BsL + row * LDH + kp * 4) = lo;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[row * LDA + kp * 4 + e] = rb[t][e];
                }
            }
        }
    };

    float4v acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int nchunks = (p.K + KB - 1) / KB;
    const int frow = lane & 15, fk = lane >> 4;
    load_regs(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_lds();
        __syncthreads();
        if (c + 1 < nchunks) load_regs((c + 1) * KB);
        if (SPLIT) {   // x w ~ xh wh + xh wl + xl wh  (the dropped xl wl is 2^-18 of the product), fp32 accumulation; smallest terms first
            bf16x8 ah[WM], al[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *(const bf16x8*)(AsH + ((wm * WM + i) * 16 + frow) * LDH + fk * 8);
                al[i] = *(const bf16x8*)(AsL + ((wm * WM + i) * 16 + frow) * LDH + fk * 8);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const bf16x8 bh = *(const bf16x8*)(BsH + ((wn * WN + j) * 16 + frow) * LDH + fk * 8);
                const bf16x8 bl = *(const bf16x8*)(BsL + ((wn * WN + j) * 16 + frow) * LDH + fk * 8);
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int ks = 0; ks < KB / 4; ++ks) {
            float af[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = As[((wm * WM + i) * 16 + frow) * LDA + ks * 4 + fk];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const float bf = Bs[((wn * WN + j) * 16 + frow) * LDA + ks * 4 + fk];
#pragma unroll
                for (int i = 0; i < WM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][j], 0, 0, 0);
            }
        }
    }
    // ---- epilogue, in HALVES rounds over half-height tiles.  The fp32 output tile in LDS (128 rows x up to 196 floats = 100 KB) used to set the
    // kernel's LDS size and with it ONE workgroup - one wave per SIMD - per CU (round 2/3: 27 - 47 TFLOP/s on the config-3 graph).  Round 4: a
    // round takes the row tiles i with i / RH == h of every wave (RH = WM / HALVES), so the tile is 64 rows x LDC: 52 KB with the row tables,
    // three workgroups per CU.  Rows are independent in everything below except the squeeze-excite column sums, which are carried across the
    // rounds per thread (same fixed order on every run).
    constexpr int RH = WM / HALVES, HROWS = WAVES_M * RH * 16;
    static_assert(WM % HALVES == 0, "row tiles per wave split evenly over the rounds");
    float pool_sum = 0.f;                                             // column tid of the tile (BN <= 256)
    float* __restrict__ Og = (float*)p.out.p;
    const int Cso = p.out.Cs;
    constexpr bool pix = PIX;
    const int ppc = (pix ? Cso : BN) / 4;                  // pieces per (row, sub-pixel) in this tile
    const int subs = pix ? BN / Cso : 1;
    constexpr int GS_ROWS = BN / 4 <= 4 ? 4 : BN / 4 <= 8 ? 8 : BN / 4 <= 16 ? 16 : BN / 4 <= 32 ? 32 : 64;
    int gs = GS_ROWS;
    if (pix) { gs = 1; while (gs < ppc) gs <<= 1; }
    const int groups = 256 / gs;
    const int jp = tid & (gs - 1);
    static_assert(HALVES <= 2, "rounds are spelled out below");
    auto round = [&](auto h_c) {                                      // (a generic lambda per round instead of an unrolled loop: the accumulator indices must be compile-time)
        constexpr int h = decltype(h_c)::value;
        __syncthreads();                                              // the operand tiles (first round) / the previous round's tile are done with
        // phase 1: accumulators -> (LayerNorm algebra, bias, activation) -> fp32 tile in LDS; the activation is chosen once per round, not once per value
        auto phase1 = [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
            const int ccol = lane & 15, crow = (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = (wn * WN + j) * 16 + ccol, n = n0 + col;
                float bias = 0.f, cs = 0.f;
                if (n < p.N) { bias = p.bias ? p.bias[n] : 0.f; if (p.ln) cs = p.csum[n]; }
#pragma unroll
                for (int ii = 0; ii < RH; ++ii)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = h * RH + ii;
                        const int row = (wm * WM + i) * 16 + crow + e, lrow = (wm * RH + ii) * 16 + crow + e;
                        float v = acc[i][j][e];
                        if (p.ln) v = s_rstd[row] * (v - s_mean[row] * cs);
                        v += bias;
                        Cs[lrow * LDC + col] = SPLIT && ACT == 2 ? gelu_poly(v) : act_fn(v, ACT < 0 ? p.act : ACT, p.alpha);
                    }
            }
        };
        if (p.act == 0) phase1(std::integral_constant<int, 0>{});
        else if (p.act == 2) phase1(std::integral_constant<int, 2>{});
        else phase1(std::integral_constant<int, -1>{});
        __syncthreads();
        // phase 2: pieces of 4 floats: residual adds, clip, store, LayerNorm statistics, SE pooling
        const int items = HROWS * subs;
        for (int q0 = 0; q0 < items; q0 += groups) {
            int q = q0 + tid / gs;
            if (q >= items) q = items - 1;   // clamp (duplicates are masked by `valid` below)
            const int li = q / subs, s = q - li * subs;
            const int i = (li / (RH * 16)) * (WM * 16) + h * (RH * 16) + li % (RH * 16);      // row of the workgroup tile behind local row li
            bool valid = (q0 + tid / gs) < items && jp < ppc && s_aoff[i] >= 0;
            const int ccol = pix ? s * Cso + jp * 4 : jp * 4;     // column inside the LDS tile
            const int ch = pix ? jp * 4 : n0 + jp * 4;            // channel inside the output pixel
            const int sg = pix ? n0 / Cso + s : 0;
            if (pix ? sg >= p.r * p.r : ch >= p.N) valid = false;
            const int dy = pix ? sg / p.r : 0, dx = pix ? sg - dy * p.r : 0;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            int b = 0, Y = 0, X = 0;
            if (valid) {
                b = s_ob[i]; Y = s_oy[i] * p.r + dy; X = s_ox[i] * p.r + dx;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Cs[li * LDC + ccol + e];
                if (p.res.p) {
                    const float4v r = *(const float4v*)((const float*)p.res.p + (size_t)((b * p.res.Hs + Y + p.res.y0) * p.res.Ws + X + p.res.x0) * p.res.Cs + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                }
                if (p.res2.p) {
                    const float4v r = *(const float4v*)((const float*)p.res2.p + (size_t)((b * p.res2.Hs + Y + p.res2.y0) * p.res2.Ws + X + p.res2.x0) * p.res2.Cs + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                }
                if (p.has_clip) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], p.clip_lo), p.clip_hi);
                }
                *(float4v*)(Og + (size_t)((b * p.out.Hs + Y) * p.out.Ws + X) * Cso + ch) = (float4v){v[0], v[1], v[2], v[3]};
                if (p.pool_out) *(float4v*)(Cs + li * LDC + ccol) = (float4v){v[0], v[1], v[2], v[3]};   // final values for the column sums below
            }
            if (p.stats_out) {   // uniform branch; all lanes take part in the shuffles
                float sum = v[0] + v[1] + v[2] + v[3];
                for (int msk = gs >> 1; msk > 0; msk >>= 1) sum += __shfl_xor(sum, msk);
                const float mean = sum / (float)p.Cout;
                float sq = 0.f;
                if (valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; sq += d * d; }
                }
                for (int msk = gs >> 1; msk > 0; msk >>= 1) sq += __shfl_xor(sq, msk);
                if (valid && jp == 0) {
                    const size_t pixi = (size_t)(b * p.out.Hs + Y) * p.out.Ws + X;
                    p.stats_out[2 * pixi] = mean;
                    p.stats_out[2 * pixi + 1] = rsqrtf(sq / (float)p.Cout + p.ln_eps);
                }
            }
        }
        if (p.pool_out) {   // per-workgroup column sums in a fixed order (the rounds' rows, in local row order); se_kernel adds the partials of a batch item in tile order
            __syncthreads();
            if (tid < BN) {
                for (int li = 0; li < HROWS; ++li) {
                    const int i = (li / (RH * 16)) * (WM * 16) + h * (RH * 16) + li % (RH * 16);
                    if (s_aoff[i] >= 0) pool_sum += Cs[li * LDC + tid];
                }
            }
        }
    };
    round(std::integral_constant<int, 0>{});
    if constexpr (HALVES > 1) round(std::integral_constant<int, 1>{});
    if (p.pool_out && tid < BN && n0 + tid < p.N) p.pool_out[(size_t)blockIdx.x * Cso + n0 + tid] = pool_sum;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SPLIT>
hipError_t launch_cfg32s(const GemmParams& p, hipStream_t s) {
    constexpr int HALVES = 2;             // the output tile goes through LDS in two half-height rounds (see the kernel's epilogue)
    constexpr int BM = WAVES_M * WM * 16, BN = WAVES_N * WN * 16, LDC = BN + 4;
    constexpr int AB = SPLIT ? (BM + BN) * 40 * 4 : (BM + BN) * 17 * 4, CB = BM / HALVES * LDC * 4;
    constexpr int SMEM = (AB > CB ? AB : CB) + BM * 6 * 4;
    static unsigned lds_ok = 0, lds_ok_pix = 0;   // per-device bits: kernels.h ensure_dynamic_lds
    const dim3 grid(p.B * ((p.Mrows + BM - 1) / BM), (p.N + BN - 1) / BN);
    if (p.omode == 2) {
        auto kern = gemm32_kernel<WAVES_M, WAVES_N, WM, WN, HALVES, SPLIT, true>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, SMEM, lds_ok_pix); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, s, p);
    } else {
        auto kern = gemm32_kernel<WAVES_M, WAVES_N, WM, WN, HALVES, SPLIT, false>;
        if (hipError_t e = ensure_dynamic_lds((const void*)kern, SMEM, lds_ok); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, s, p);
    }
    return hipGetLastError();
}
template <int WAVES_M, int WAVES_N, int WM, int WN>
hipError_t launch_cfg32(const GemmParams& p, hipStream_t s, bool exact) {
    return exact ? launch_cfg32s<WAVES_M, WAVES_N, WM, WN, false>(p, s) : launch_cfg32s<WAVES_M, WAVES_N, WM, WN, true>(p, s);
}

// ---- window attention core, fp32 rows ------------------------------------------------------------------------------------------
template <int HD, int NTOK>
__global__ __launch_bounds__(256) void attn32_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float sK[4][NTOK][HD];
    __shared__ __attribute__((aligned(16))) float sV[4][NTOK][HD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int C = p.heads * HD;
    const long total = (long)p.B * p.nwin * p.heads;
    long unit = (long)blockIdx.x * 4 + wv;
    const bool active_wave = unit < total;
    if (!active_wave) unit = total - 1;   // keep the wave alive for the barrier; results are not stored
    const int h = (int)(unit % p.heads);
    const long win = unit / p.heads;           // global window index (b * nwin + w)
    const int w = (int)(win % p.nwin);
    const bool act = lane < NTOK;
    const float* base = (const float*)p.qkv + (win * NTOK + (act ? lane : 0)) * (long)(3 * C) + h * HD;

    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4v qv = *(const float4v*)(base + d), kv = *(const float4v*)(base + C + d), vv = *(const float4v*)(base + 2 * C + d);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[d + e] = qv[e] * p.scale;
        if (act) { *(float4v*)&sK[wv][lane][d] = kv; *(float4v*)&sV[wv][lane][d] = vv; }
    }
    __syncthreads();

    const float* bias = (const float*)p.bias + (((long)p.maskid[w] * p.heads + h) * NTOK + (act ? lane : 0)) * NTOK;
    float s[NTOK];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4v kv = *(const float4v*)&sK[wv][j][d];
#pragma unroll
            for (int e = 0; e < 4; ++e) a += q[d + e] * kv[e];
        }
        a += bias[j];
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    const float inv = 1.f / sum;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
        const float pj = s[j];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4v vv = *(const float4v*)&sV[wv][j][d];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[d + e] += pj * vv[e];
        }
    }
    if (act && active_wave) {
        float* op_ = (float*)p.out + (win * NTOK + lane) * (long)C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *(float4v*)(op_ + d) = (float4v){o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
    }
}

// ---- fused transformer MLP branch on fp32 rows, Precision::TF32 (round 6) ---------------------------------------------------------------
//     y = x + W2 gelu(W1 LayerNorm(x) + b1) + b2          rows [M][C] fp32, C = 96 / 192, hidden 2C
// The un-fused fp32 plan runs this as two gemm32 launches with the hidden map [M][2C] in HBM between them: 2.1 GB written and read again per full-resolution
// block of config 3, more than x and y together (1.06 GB each) - fc1 + fc2 were 19.5 of the frame's 49.8 ms.  Here a workgroup of six waves owns 64 rows:
//   * the rows are normalised with the producer's statistics (mean, rstd per row: the plan's stats tensor), split x = hi + lo into two bf16 planes (split4)
//     and stay in LDS for the first product;
//   * the hidden units go by in chunks of 96: wave w computes units 16 w .. 16 w + 15 of the chunk for all 64 rows - transposed (rows = hidden units,
//     columns = tokens), three v_mfma_f32_16x16x32_bf16 per k-step as in gemm32_kernel (lo hi + hi lo + hi hi) - adds b1, applies GELU (gelu_poly) and writes
//     the result, split again, as 8-byte pieces into the chunk's hidden tile in LDS (a lane holds four consecutive units of one token);
//   * the second product accumulates out^T (rows = output channels, columns = tokens) over the chunks, wave w owning C / 6 output channels; a lane ends with
//     four consecutive channels of a token: + b2 + x (read again, a cache hit), 16-byte stores;
//   * weights never enter LDS: the host keeps both matrices as bf16 hi / lo planes in fragment-major order (fragorder.h frag_major), a fragment is one
//     coalesced KiB per wave, requested a whole chunk ahead;
//   * LayerNorm statistics of the produced rows (for an un-fused consumer): two passes like gemm32_kernel's epilogue (mean, then the squared distances),
//     per-wave partial sums through LDS added in wave order - the same bits on every run.
// Precision::FP32 (exact products) keeps the two launches.
template <int C>
__global__ __launch_bounds__(384, 3) void mlp32_kernel(const Mlp32Params p) {
    constexpr int BM = 64, TT = BM / 16, HC = 96, NCH = 2 * C / HC, KS1 = C / 32, KS2 = HC / 32, NT2 = C / HC;
    constexpr int LDX = C + 8, LDHH = HC + 8;                        // bf16 per row: 16-byte aligned rows that rotate over the banks
    static_assert(C % HC == 0 && (2 * C) % HC == 0, "widths in chunks of 96");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* XH = (unsigned short*)smem;
    unsigned short* XL = XH + BM * LDX;
    unsigned short* HH = XL + BM * LDX;
    unsigned short* HL = HH + BM * LDHH;
    float* red = (float*)smem;                                        // [6 waves][BM] partial row sums; over the x planes, which are done with by then

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const long row0 = (long)blockIdx.x * BM;

    // ---- rows in: normalise, split, two planes in LDS (rows past the end: zeros)
    for (int idx = tid; idx < BM * (C / 4); idx += 384) {
        const int row = idx / (C / 4), c4 = idx - row * (C / 4);
        const long r = row0 + row;
        float4v v = {0.f, 0.f, 0.f, 0.f};
        float mean = 0.f, rstd = 0.f;
        if (r < p.M) { v = *(const float4v*)(p.x + r * C + c4 * 4); mean = p.stats_in[2 * r]; rstd = p.stats_in[2 * r + 1]; }
        uint2v hi, lo;
        split4((float4v){(v[0] - mean) * rstd, (v[1] - mean) * rstd, (v[2] - mean) * rstd, (v[3] - mean) * rstd}, hi, lo);
        *(uint2v*)(XH + row * LDX + c4 * 4) = hi;
        *(uint2v*)(XL + row * LDX + c4 * 4) = lo;
    }
    const bf16x8* W1H = (const bf16x8*)p.w1h + lane;                  // [2C / 16 row tiles][KS1][64 lanes][8]
    const bf16x8* W1L = (const bf16x8*)p.w1l + lane;
    const bf16x8* W2H = (const bf16x8*)p.w2h + lane;                  // [C / 16 row tiles][2C / 32 k-steps][64 lanes][8]
    const bf16x8* W2L = (const bf16x8*)p.w2l + lane;
    float4v acc2[NT2][TT];
#pragma unroll
    for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc2[n][tt] = (float4v){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

#pragma unroll 1
    for (int hc = 0; hc < NCH; ++hc) {
        // ---- first product, this wave's 16 hidden units of the chunk: weights requested for the whole chunk at once
        const int ht = hc * (HC / 16) + wv;                           // hidden row tile
        bf16x8 wh[KS1], wl[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) { wh[ks] = W1H[(size_t)(ht * KS1 + ks) * 64]; wl[ks] = W1L[(size_t)(ht * KS1 + ks) * 64]; }
        // (the second product's fragments of this chunk too: they travel under the first product and the GELU)
        bf16x8 vh[NT2][KS2], vl[NT2][KS2];
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const size_t f = (size_t)((wv * NT2 + n) * (2 * C / 32) + hc * KS2 + ks) * 64;
                vh[n][ks] = W2H[f]; vl[n][ks] = W2L[f];
            }
        float4v acc1[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc1[tt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const bf16x8 xh = *(const bf16x8*)(XH + (tt * 16 + frow) * LDX + ks * 32 + fk * 8);
                const bf16x8 xl = *(const bf16x8*)(XL + (tt * 16 + frow) * LDX + ks * 32 + fk * 8);
                acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], xh, acc1[tt], 0, 0, 0);
                acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], xl, acc1[tt], 0, 0, 0);
                acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], xh, acc1[tt], 0, 0, 0);
            }
        // ---- + b1, GELU, split: a lane holds hidden units 16 ht + 4 fk .. + 3 of token 16 tt + frow
        const float4v b1v = *(const float4v*)(p.b1 + ht * 16 + fk * 4);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float4v h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = gelu_poly(acc1[tt][e] + b1v[e]);
            uint2v hi, lo;
            split4(h, hi, lo);
            *(uint2v*)(HH + (tt * 16 + frow) * LDHH + wv * 16 + fk * 4) = hi;
            *(uint2v*)(HL + (tt * 16 + frow) * LDHH + wv * 16 + fk * 4) = lo;
        }
        __syncthreads();
        // ---- second product over the chunk's 96 hidden units: this wave's output channels
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const bf16x8 hh = *(const bf16x8*)(HH + (tt * 16 + frow) * LDHH + ks * 32 + fk * 8);
                const bf16x8 hl = *(const bf16x8*)(HL + (tt * 16 + frow) * LDHH + ks * 32 + fk * 8);
#pragma unroll
                for (int n = 0; n < NT2; ++n) {
                    acc2[n][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[n][ks], hh, acc2[n][tt], 0, 0, 0);
                    acc2[n][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[n][ks], hl, acc2[n][tt], 0, 0, 0);
                    acc2[n][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[n][ks], hh, acc2[n][tt], 0, 0, 0);
                }
            }
        __syncthreads();                                              // the next chunk's hidden tile goes over this one (after the last chunk: the planes are free)
    }

    // ---- + b2 + x, stores: a lane holds output channels 16 (wv NT2 + n) + 4 fk .. + 3 of token 16 tt + frow
    float rsum[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) rsum[tt] = 0.f;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int ch = (wv * NT2 + n) * 16 + fk * 4;
        const float4v b2v = *(const float4v*)(p.b2 + ch);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const long r = row0 + tt * 16 + frow;
            float4v v = acc2[n][tt] + b2v;
            if (r < p.M) {
                v += *(const float4v*)(p.x + r * C + ch);
                *(float4v*)(p.y + r * C + ch) = v;
            } else v = (float4v){0.f, 0.f, 0.f, 0.f};
            acc2[n][tt] = v;
            rsum[tt] += v[0] + v[1] + v[2] + v[3];
        }
    }
    if (p.stats_out) {                                                // uniform
        auto reduce_rows = [&](float (&part)[TT], float (&total)[TT]) {   // sums over the four lane groups, then over the six waves in wave order
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                float v = part[tt];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (fk == 0) red[wv * BM + tt * 16 + frow] = v;
            }
            __syncthreads();
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 6; ++w) t += red[w * BM + tt * 16 + frow];
                total[tt] = t;
            }
            __syncthreads();
        };
        float tot[TT], mean[TT], sq[TT];
        reduce_rows(rsum, tot);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            mean[tt] = tot[tt] / (float)C;
            float q = 0.f;
#pragma unroll
            for (int n = 0; n < NT2; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = acc2[n][tt][e] - mean[tt]; q += d * d; }
            sq[tt] = q;
        }
        reduce_rows(sq, tot);
        if (wv == 0 && fk == 0) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const long r = row0 + tt * 16 + frow;
                if (r < p.M) { p.stats_out[2 * r] = mean[tt]; p.stats_out[2 * r + 1] = rsqrtf(tot[tt] / (float)C + p.eps_out); }
            }
        }
    }
}

template <int C>
hipError_t launch_mlp32_c(const Mlp32Params& p, hipStream_t s) {
    constexpr int SMEM = 64 * (C + 8) * 2 * 2 + 64 * (96 + 8) * 2 * 2;
    static_assert(SMEM >= 6 * 64 * 4, "the statistics' partial sums fit the x planes");
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp32_kernel<C>, SMEM, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(mlp32_kernel<C>, dim3((unsigned)((p.M + 63) / 64)), dim3(384), SMEM, s, p);
    return hipGetLastError();
}

// ---- fused Swin attention branch on fp32 rows, Precision::TF32 (round 6) -------------------------------------------------------------
//     y = x + proj( W-MSA( LayerNorm(x) ) )          windows of 6 x 6 tokens, six heads of 16 (C = 96) or 32 (C = 192)
// The un-fused fp32 plan runs this as gemm32 (LayerNorm + window gather + qkv) -> attn32 (lane-per-query core) -> gemm32 (proj + window scatter + residual) with the
// qkv map [M][3C] and the head outputs [M][C] in HBM between them: 8.5 GB of traffic per full-resolution block of config 3 for 2.1 GB of rows in and out.  Here a
// workgroup of six waves (wave = head) owns two or three windows; k_swinattn192u.hip's recipe per (window, head) unit with every product as three bf16 products
// (lo hi + hi lo + hi hi, split4):
//   * the windows' rows are gathered (the qkv op's window table), normalised with the producer's statistics, split and kept in LDS as two bf16 planes of 48 rows
//     per window (rows 36 .. 47 zero);
//   * q^T and k^T are computed transposed (rows = features: A = weights, B = x), v plain (A = x, B = weights): their accumulators are, as they stand, the operands of
//     S^T = K Q^T (k order = the features as a lane holds them) and of O^T = V^T P^T (k order = the keys as the S^T accumulators hold them); q bias = initial
//     accumulator, k bias dropped (constant per query), v bias added after the normalisation;
//   * scores start from the relative-position bias (+ shift mask) table, softmax over the keys is lane-local plus two lane swaps, padded keys are -inf;
//   * the head outputs go back to LDS - over the rows of the window they were computed from, behind a barrier - as the two planes the projection reads;
//   * proj transposed (rows = output channels, wave w owns C / 6 of them), + bias + residual, scattered through the proj op's window table; LayerNorm statistics of
//     the produced rows like mlp32_kernel.
// Weights: bf16 hi / lo planes in fragment-major order, streamed from L2 per window.
struct Bf2 { bf16x8 h, l; };
__device__ __forceinline__ Bf2 split8(const float4v a, const float4v b) {
    uint2v ah, al, bh, bl;
    split4(a, ah, al); split4(b, bh, bl);
    typedef unsigned uint4v_ __attribute__((ext_vector_type(4)));
    Bf2 r;
    r.h = __builtin_bit_cast(bf16x8, (uint4v_){ah[0], ah[1], bh[0], bh[1]});
    r.l = __builtin_bit_cast(bf16x8, (uint4v_){al[0], al[1], bl[0], bl[1]});
    return r;
}
__device__ __forceinline__ float4v mfma3(const Bf2& a, const Bf2& b, float4v acc) {   // a b ~ al bh + ah bl + ah bh, smallest terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0);
}

#ifndef W2X_A32_NW96
#define W2X_A32_NW96 3       // windows per workgroup at C = 96 (tools/ab/lib_variants.sh "k_f32.hip:-DW2X_A32_NW96=2")
#endif
template <int C, int HD, int NW>
__global__ __launch_bounds__(384, 2) void swinattn32_kernel(const SwinAttn32Params p) {
    constexpr int NTOK = 36, SLAB = 48, ROWS = NW * SLAB, RT = ROWS / 16, LDX = C + 8, KS = C / 32, DT = HD / 16, NH = 6, NT2 = C / 96;
    static_assert(C == NH * HD && (HD == 16 || HD == 32), "six heads of 16 or 32");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* XH = (unsigned short*)smem;                       // [ROWS][LDX]; a window's rows become its head outputs once every head has its q, k, v
    unsigned short* XL = XH + ROWS * LDX;
    int* pixi = (int*)(XL + ROWS * LDX);                              // [ROWS] pixel of the row on the input side (-1: none)
    int* pixo = pixi + ROWS;                                          // [ROWS] pixel the row is stored to
    float* red = (float*)(pixo + ROWS);                               // [6][ROWS] partial row sums of the statistics

    const int tid = threadIdx.x, lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = head (and, in the projection, a sixth of the output channels)
    const int frow = lane & 15, fk = lane >> 4;
    const long W0 = (long)blockIdx.x * NW, total = (long)p.B * p.nwin;
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int r = tid; r < ROWS; r += 384) {
        const int wi = r / SLAB, t = r - wi * SLAB;
        const long gw = W0 + wi;
        int pin = -1, pout = -1;
        if (t < NTOK && gw < total) {
            const long b = gw / p.nwin; const int wl = (int)(gw - b * p.nwin);
            pin = (int)(b * p.pix_per_item + p.table_in[wl * NTOK + t]);
            pout = (int)(b * p.pix_per_item + p.table_out[wl * NTOK + t]);
        }
        pixi[r] = pin; pixo[r] = pout;
    }
    __syncthreads();
    for (int idx = tid; idx < ROWS * (C / 4); idx += 384) {
        const int row = idx / (C / 4), c4 = idx - row * (C / 4);
        const int pin = pixi[row];
        float4v v = zero4;
        float mean = 0.f, rstd = 0.f;
        if (pin >= 0) { v = *(const float4v*)(p.x + (size_t)pin * C + c4 * 4); mean = p.stats_in[2 * (size_t)pin]; rstd = p.stats_in[2 * (size_t)pin + 1]; }
        uint2v hi, lo;
        split4((float4v){(v[0] - mean) * rstd, (v[1] - mean) * rstd, (v[2] - mean) * rstd, (v[3] - mean) * rstd}, hi, lo);
        *(uint2v*)(XH + row * LDX + c4 * 4) = hi;
        *(uint2v*)(XL + row * LDX + c4 * 4) = lo;
    }
    __syncthreads();

    // weight planes through buffer resources: the lane's 16 bytes in one VGPR for every request, the fragment's KiB in the scalar offset (64-bit per-lane addresses
    // cost two registers per request in flight)
    const __amdgpu_buffer_rsrc_t WQH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wqkv_h), 0, 3u * C * C * 2u, 0x00020000);   // [3C / 16 row tiles][KS][64 lanes][8]
    const __amdgpu_buffer_rsrc_t WQL = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wqkv_l), 0, 3u * C * C * 2u, 0x00020000);
    const unsigned l16 = lane * 16u;
    auto wfrag = [&](int sel, int dt, int ks) {                       // fragment of matrix sel (0 q, 1 k, 2 v), feature tile dt of this head, k-step ks
        const unsigned f = (unsigned)(((sel * C + h * HD) / 16 + dt) * KS + ks) * 1024u;
        Bf2 w;
        w.h = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(WQH, l16, f, 0));
        w.l = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(WQL, l16, f, 0));
        return w;
    };
    const float qs = p.scale;
    // AHEAD (C = 96, where the register budget of three waves per SIMD has room): the first fragments of every phase are requested a phase early - q / k of the next
    // window under this window's v products and softmax (a head's first k-step is the same for every window), v under the last q / k step, proj before the windows -
    // instead of at the top of the loop that needs them (five exposed L2 round trips per workgroup)
    constexpr bool AHEAD = C == 96;
    Bf2 wq0[DT], wk0[DT], wv0[DT];
    if constexpr (AHEAD) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { wq0[dt] = wfrag(0, dt, 0); wk0[dt] = wfrag(1, dt, 0); }
    }

#pragma unroll 1
    for (int wi = 0; wi < NW; ++wi) {
        const long gw = W0 + wi;
        const bool wok = gw < total;
        const int wl = wok ? (int)(gw % p.nwin) : 0;
        const unsigned short* xh = XH + wi * SLAB * LDX;
        const unsigned short* xl = XL + wi * SLAB * LDX;
        // ---- q^T, k^T (rows = features, columns = tokens), then v (rows = tokens, columns = features).  Weight fragments are requested one k-step ahead and the
        //      requests are pinned where they are written (left alone the compiler asks for every k-step of every matrix first: 50 spilled registers at C = 192)
        float4v aq[DT][3], ak[DT][3], av[3][DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float4v bq = *(const float4v*)(p.bqkv + h * HD + dt * 16 + fk * 4);
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) { aq[dt][tt] = bq; ak[dt][tt] = zero4; }
        }
        auto xfrags = [&](int ks, Bf2 (&x)[3]) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                x[tt].h = *(const bf16x8*)(xh + (tt * 16 + frow) * LDX + ks * 32 + fk * 8);
                x[tt].l = *(const bf16x8*)(xl + (tt * 16 + frow) * LDX + ks * 32 + fk * 8);
            }
        };
        {
            Bf2 wq[2][DT], wk[2][DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if constexpr (AHEAD) { wq[0][dt] = wq0[dt]; wk[0][dt] = wk0[dt]; } else { wq[0][dt] = wfrag(0, dt, 0); wk[0][dt] = wfrag(1, dt, 0); }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { wq[(ks + 1) & 1][dt] = wfrag(0, dt, ks + 1); wk[(ks + 1) & 1][dt] = wfrag(1, dt, ks + 1); }
                } else if constexpr (AHEAD) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) wv0[dt] = wfrag(2, dt, 0);
                }
                asm volatile("" ::: "memory");
                Bf2 x[3];
                xfrags(ks, x);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) {
                        aq[dt][tt] = mfma3(wq[ks & 1][dt], x[tt], aq[dt][tt]);
                        ak[dt][tt] = mfma3(wk[ks & 1][dt], x[tt], ak[dt][tt]);
                    }
            }
        }
        {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) av[tt][dt] = zero4;
            Bf2 wv[2][DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { if constexpr (AHEAD) wv[0][dt] = wv0[dt]; else wv[0][dt] = wfrag(2, dt, 0); }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) wv[(ks + 1) & 1][dt] = wfrag(2, dt, ks + 1);
                } else if constexpr (AHEAD) {      // the next window's first q / k fragments (the same ones: requested again, landing under the softmax)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { wq0[dt] = wfrag(0, dt, 0); wk0[dt] = wfrag(1, dt, 0); }
                }
                asm volatile("" ::: "memory");
                Bf2 x[3];
                xfrags(ks, x);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) av[tt][dt] = mfma3(x[tt], wv[ks & 1][dt], av[tt][dt]);
            }
        }
        // the accumulators as operands: q / k tile tt = [features of tile 0 | features of tile 1] of 16 tokens; v: keys of tiles 0 | 1, and of tile 2 | nothing
        Bf2 qf[3], kf[3], vf0[DT], vf1[DT];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            qf[tt] = split8(aq[0][tt] * qs, DT > 1 ? aq[DT - 1][tt] * qs : zero4);
            kf[tt] = split8(ak[0][tt], DT > 1 ? ak[DT - 1][tt] : zero4);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { vf0[dt] = split8(av[0][dt], av[1][dt]); vf1[dt] = split8(av[2][dt], zero4); }
        __syncthreads();                                              // every head has read this window's rows: the head outputs may go over them

        // ---- per query tile: S^T = K Q^T on top of the bias (a lane holds keys 16 kt + 4 fk + e of query 16 qi + frow), softmax over the keys, O^T = V^T P^T
        //      normalised + v bias (a lane holds features 16 dt + 4 fk + e of the query) -> the window's rows in LDS, split again.  One query tile at a time:
        //      scores and probabilities of all three at once cost 56 more registers (47 spilled at C = 192)
        const float* bias = p.bias + ((size_t)(wok ? p.maskid[wl] : 0) * NH + h) * (NTOK * NTOK);
        float4v bv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) bv[dt] = *(const float4v*)(p.bqkv + 2 * C + h * HD + dt * 16 + fk * 4);
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            const int q = qi * 16 + frow;
            float4v sc[3];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const int k0 = kt * 16 + fk * 4;
                float4v b = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if (k0 < NTOK) b = q < NTOK ? *(const float4v*)(bias + q * NTOK + k0) : zero4;
                sc[kt] = mfma3(kf[kt], qf[qi], b);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sc[kt][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = __expf(sc[kt][e] - mx); sc[kt][e] = v; sum += v; }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.f / sum;
            const Bf2 pf0 = split8(sc[0], sc[1]), pf1 = split8(sc[2], zero4);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float4v o = mfma3(vf0[dt], pf0, zero4);
                o = mfma3(vf1[dt], pf1, o);
                uint2v hi, lo;
                split4(o * inv + bv[dt], hi, lo);
                const int row = wi * SLAB + qi * 16 + frow, col = h * HD + dt * 16 + fk * 4;
                *(uint2v*)(XH + row * LDX + col) = hi;
                *(uint2v*)(XL + row * LDX + col) = lo;
            }
            asm volatile("" ::: "memory");                            // (one query tile after the other)
        }
    }
    __syncthreads();

    // ---- proj, transposed: this wave's output channels for every row tile
    const __amdgpu_buffer_rsrc_t WPH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wproj_h), 0, (unsigned)C * C * 2u, 0x00020000);   // [C / 16 row tiles][KS][64 lanes][8]
    const __amdgpu_buffer_rsrc_t WPL = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wproj_l), 0, (unsigned)C * C * 2u, 0x00020000);
    float4v acc[NT2][RT];
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const float4v bp = *(const float4v*)(p.bproj + (h * NT2 + n) * 16 + fk * 4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[n][rt] = bp;
    }
    auto pfrag = [&](int n, int ks) {
        const unsigned f = (unsigned)((h * NT2 + n) * KS + ks) * 1024u;
        Bf2 w;
        w.h = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(WPH, l16, f, 0));
        w.l = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(WPL, l16, f, 0));
        return w;
    };
    Bf2 wpb[2][NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n) wpb[0][n] = pfrag(n, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
            for (int n = 0; n < NT2; ++n) wpb[(ks + 1) & 1][n] = pfrag(n, ks + 1);
        }
        asm volatile("" ::: "memory");
        const Bf2 (&wp)[NT2] = wpb[ks & 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            Bf2 o;
            o.h = *(const bf16x8*)(XH + (rt * 16 + frow) * LDX + ks * 32 + fk * 8);
            o.l = *(const bf16x8*)(XL + (rt * 16 + frow) * LDX + ks * 32 + fk * 8);
#pragma unroll
            for (int n = 0; n < NT2; ++n) acc[n][rt] = mfma3(wp[n], o, acc[n][rt]);
        }
    }
    // ---- + residual, scatter store: a lane holds output channels 16 (h NT2 + n) + 4 fk + e of row 16 rt + frow
    float rsum[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) rsum[rt] = 0.f;
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int ch = (h * NT2 + n) * 16 + fk * 4;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int pout = pixo[rt * 16 + frow];
            float4v v = acc[n][rt];
            if (pout >= 0) {
                v += *(const float4v*)(p.res + (size_t)pout * C + ch);
                *(float4v*)(p.y + (size_t)pout * C + ch) = v;
            } else v = zero4;
            acc[n][rt] = v;
            rsum[rt] += v[0] + v[1] + v[2] + v[3];
        }
    }
    if (p.stats_out) {                                                // uniform
        auto reduce_rows = [&](float (&part)[RT], float (&total_)[RT]) {   // sums over the four lane groups, then over the six waves in wave order
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float v = part[rt];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (fk == 0) red[h * ROWS + rt * 16 + frow] = v;
            }
            __syncthreads();
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 6; ++w) t += red[w * ROWS + rt * 16 + frow];
                total_[rt] = t;
            }
            __syncthreads();
        };
        float tot[RT], mean[RT], sq[RT];
        reduce_rows(rsum, tot);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            mean[rt] = tot[rt] / (float)C;
            float q = 0.f;
#pragma unroll
            for (int n = 0; n < NT2; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = acc[n][rt][e] - mean[rt]; q += d * d; }
            sq[rt] = pixo[rt * 16 + frow] >= 0 ? q : 0.f;
        }
        reduce_rows(sq, tot);
        if (h == 0 && fk == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int pout = pixo[rt * 16 + frow];
                if (pout >= 0) { p.stats_out[2 * (size_t)pout] = mean[rt]; p.stats_out[2 * (size_t)pout + 1] = rsqrtf(tot[rt] / (float)C + p.eps_out); }
            }
        }
    }
}

template <int C, int HD, int NW>
hipError_t launch_swinattn32_c(const SwinAttn32Params& p, hipStream_t s) {
    constexpr int ROWS = NW * 48, SMEM = ROWS * (C + 8) * 2 * 2 + ROWS * 4 * 2 + 6 * ROWS * 4;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)swinattn32_kernel<C, HD, NW>, SMEM, lds_ok); e != hipSuccess) return e;
    const long total = (long)p.B * p.nwin;
    hipLaunchKernelGGL((swinattn32_kernel<C, HD, NW>), dim3((unsigned)((total + NW - 1) / NW)), dim3(384), SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

bool mlp32_supported(int C) { return C == 96 || C == 192; }
hipError_t launch_mlp32(const Mlp32Params& p, hipStream_t s) {
    if (p.M <= 0 || !p.x || !p.y || !p.stats_in || !p.w1h || !p.w1l || !p.w2h || !p.w2l || !p.b1 || !p.b2) return hipErrorInvalidValue;
    if (p.C == 96) return launch_mlp32_c<96>(p, s);
    if (p.C == 192) return launch_mlp32_c<192>(p, s);
    return hipErrorInvalidValue;
}

bool swinattn32_supported(int C, int heads, int hd, int ntok) { return heads == 6 && ntok == 36 && ((C == 96 && hd == 16) || (C == 192 && hd == 32)); }
hipError_t launch_swinattn32(const SwinAttn32Params& p, hipStream_t s) {
    if (p.B <= 0 || p.nwin <= 0 || !p.x || !p.y || !p.res || !p.stats_in || !p.table_in || !p.table_out || !p.wqkv_h || !p.wqkv_l || !p.wproj_h || !p.wproj_l ||
        !p.bqkv || !p.bproj || !p.bias || !p.maskid) return hipErrorInvalidValue;
    if ((long)p.B * p.pix_per_item > 0x7FFFFFFFl) return hipErrorInvalidValue;        // pixel indices are ints in the kernel's tables
    // windows per workgroup: three at C = 96 (60 KB of row planes: two workgroups per CU), two at C = 192 (77 KB)
    if (p.C == 96) return launch_swinattn32_c<96, 16, W2X_A32_NW96>(p, s);
    if (p.C == 192) return launch_swinattn32_c<192, 32, 2>(p, s);
    return hipErrorInvalidValue;
}

// Tile selection as in launch_gemm (k_gemm.hip): the tile width follows N; pixel-shuffle outputs take whole output pixels.
hipError_t launch_gemm_f32(const GemmParams& p, hipStream_t s, bool exact) {
    if (p.a.Cs % 4 || p.out.Cs % 4 || p.Kw % 4 || p.K % 4) return hipErrorInvalidValue;   // pieces of four floats
    int bn;
    if (p.omode == 2) bn = (p.out.Cs <= 192 && 192 % p.out.Cs == 0 && p.N >= 192) ? 192 : p.out.Cs;
    else bn = p.N % 192 == 0 ? 192 : p.N % 96 == 0 ? 96 : p.N % 128 == 0 ? 128 : p.N % 64 == 0 ? 64 : p.N % 48 == 0 ? 48 : p.N % 32 == 0 ? 32 : p.N <= 16 ? 16 : 0;
    if (p.omode == 2 && bn < 16) bn = 16;                           // 4 sub-pixels x 4 stored channels and the like: whole tile
    if (p.stats_out && p.omode != 2 && bn < p.N) return hipErrorInvalidValue;
    if (p.omode == 2 && (bn % p.out.Cs)) return hipErrorInvalidValue;
    switch (bn) {
        case 192: return launch_cfg32<2, 2, 4, 6>(p, s, exact);
        case 128: return launch_cfg32<2, 2, 4, 4>(p, s, exact);
        case 96: return launch_cfg32<4, 1, 2, 6>(p, s, exact);
        case 64: return launch_cfg32<4, 1, 2, 4>(p, s, exact);
        case 48: return launch_cfg32<4, 1, 2, 3>(p, s, exact);
        case 32: return launch_cfg32<4, 1, 2, 2>(p, s, exact);
        case 16: return launch_cfg32<4, 1, 2, 1>(p, s, exact);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_attn_f32(const AttnParams& p, hipStream_t s) {
    const long total = (long)p.B * p.nwin * p.heads;
    const dim3 grid((unsigned)((total + 3) / 4));
#define W2X_ATTN32_CASE(HD_, NTOK_)                                                                                     \
    if (p.ntok == NTOK_ && p.hd == HD_) { hipLaunchKernelGGL((attn32_kernel<HD_, NTOK_>), grid, dim3(256), 0, s, p); return hipGetLastError(); }
    W2X_ATTN32_CASE(16, 36) W2X_ATTN32_CASE(32, 36) W2X_ATTN32_CASE(8, 36) W2X_ATTN32_CASE(16, 64) W2X_ATTN32_CASE(32, 64)
#undef W2X_ATTN32_CASE
    return hipErrorInvalidValue;
}

}  // namespace w2x
