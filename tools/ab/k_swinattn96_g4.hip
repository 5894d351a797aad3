// Experiment kept for the record (tools/ab/attn_ab.hip can time it): "a wave is a head" with 6-wave workgroups of 4 windows.  28 % fewer
// instructions per token than the round-1 schedule, yet 0.74-0.80 ms against 0.61 ms per launch: a 6-wave workgroup lands 2,2,1,1 on the
// four SIMDs of a CU, the second workgroup of the CU does not complement it (at 168 VGPRs it does not fit at all: half occupancy; at 128
// VGPRs the SIMDs carry 4,4,2,2 waves and the barriers wait for the loaded pair).  Workgroups must be a multiple of 4 waves.
// Fused Swin attention branch, C = 96 / 6 heads of 16 / window 6x6 for gfx950:  y = x + proj( W-MSA( LayerNorm(x) ) )
// (LayerNormalization, roll + window partition, QKV MatMul+Add, per-head scale / QK^T / rel-pos bias (+ shift mask) / Softmax /
//  .V, head merge, proj MatMul+Add, window reverse + reverse roll, residual Add of the ONNX graph: one launch.)
//
// Schedule ("a wave is a head"): a workgroup = 6 waves = 4 windows (144 token rows = 9 row tiles, no padding in the row-wise
// phases).  Wave h keeps the q/k/v weight fragments of head h in registers for the whole kernel (36 VGPRs, read once from the
// fragment-major copy) and walks the four windows with them; q, k, v, S and P never leave registers: with a head dimension of
// 16 the accumulator layout (lane (col, g) holds rows 4g..4g+3) is the operand layout of v_mfma_f32_16x16x16_f16, so
//     q^T, k^T (rows = features)  -> B / A operands of S^T = K Q^T,
//     v (rows = tokens)           -> A operand of O^T = V^T P^T,
//     S^T after the softmax       -> B operand of O^T.
// A window's 36 tokens are two full 16-token tiles plus 4 left over.  Left-over KEYS sit on rows 0,4,8,12 of a third key tile
// (slab rows 32,36,40,44), so a lane holds exactly one of them.  Left-over QUERIES of the four windows share ONE query tile
// (column 4w+i = query 32+i of window w): their q comes from one product over a cross-window fragment gather, their scores are
// accumulated over the four windows with the other windows' columns of q zeroed (S_left += K_w Qz_w), and they take a single
// softmax pass at the end instead of one mostly empty pass per window and head.
// The kernel is instruction-issue bound (SQ counters: three waves per SIMD, each ~31 % issuing), so the design minimises
// instructions per token: no padded row passes (24 rows x 16 lanes per pass, 6 passes), rows that do not exist are loaded from
// a zero page instead of being masked, LayerNorm scaling on v_fma_mix (f16 in, f32 math, f16 out: one instruction per element),
// softmax denominators from a ones-operand MFMA, k bias dropped / v bias after the normalisation, proj computed transposed so
// that a lane ends with 4 consecutive channels of one token (bias as the initial accumulator, 8-byte LDS stores).
#include "kernels.h"

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) unsigned kZeroPageG4[4] = {0u, 0u, 0u, 0u};   // source of rows that do not exist

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
// sum over aligned groups of 16 lanes with DPP
__device__ __forceinline__ float group_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}
// Row-group sums of six independent values at once: v_add_f32 with a DPP operand; the chains are interleaved so that each
// one's two wait states between a VALU write and a DPP read are filled by the others.
#define W2X_DPP1(R, CTRL) "v_add_f32_dpp " R ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define W2X_DPP6(CTRL) W2X_DPP1("%0", CTRL) W2X_DPP1("%1", CTRL) W2X_DPP1("%2", CTRL) W2X_DPP1("%3", CTRL) W2X_DPP1("%4", CTRL) W2X_DPP1("%5", CTRL)
__device__ __forceinline__ void group_sum16_x6(float& a, float& b, float& c, float& d, float& e, float& f) {
    asm volatile("s_nop 1\n\t" W2X_DPP6("quad_perm:[1,0,3,2]") W2X_DPP6("quad_perm:[2,3,0,1]") W2X_DPP6("row_half_mirror") W2X_DPP6("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
// Maximum over the four lanes that hold one query column (lanes fr, fr+16, fr+32, fr+48) for two / three independent values.
// Wait states of v_permlane*_swap (two after the VALU write it reads, one before a VALU reads its result) are filled by the
// other chain(s) or an s_nop; v_max_f32 as is (fmaxf() would canonicalise both swap results first).
__device__ __forceinline__ void cols_max2(float& a0, float& a1) {
    float b0, b1;
    asm volatile(
        "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\t"
        "v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\t"
        "v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3\n\t"
        "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
        "v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
        : "+v"(a0), "+v"(a1), "=&v"(b0), "=&v"(b1));
}
__device__ __forceinline__ void cols_max3(float& a0, float& a1, float& a2) {
    float b0, b1, b2;
    asm volatile(
        "v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\t"
        "v_permlane16_swap_b32 %0, %3\n\tv_permlane16_swap_b32 %1, %4\n\tv_permlane16_swap_b32 %2, %5\n\t"
        "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5\n\t"
        "v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\t"
        "v_permlane32_swap_b32 %0, %3\n\tv_permlane32_swap_b32 %1, %4\n\tv_permlane32_swap_b32 %2, %5\n\t"
        "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5"
        : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(b0), "=&v"(b1), "=&v"(b2));
}
// max of the nine scores a lane holds for one query (two key tiles of four, one left-over key).  Plain fmaxf on purpose: the
// inputs are MFMA results, and the wait states between an MFMA and a reader of its result are only inserted for instructions
// the compiler sees (an inline-asm v_max3 here read stale accumulators now and then: results off by an LSB from run to run).
// The chain compiles to four v_max3_f32.
__device__ __forceinline__ float max9(const float4v a, const float4v b, float c) {
    return fmaxf(fmaxf(fmaxf(fmaxf(a[0], a[1]), a[2]), fmaxf(fmaxf(a[3], b[0]), b[1])), fmaxf(fmaxf(b[2], b[3]), c));
}

constexpr int C = 96, HD = 16, NH = 6, NTOK = 36, G = 4, R = G * NTOK, RT = R / 16;   // 144 rows = 9 row tiles
constexpr int NTHR = 64 * NH;              // 384 threads: wave = head
constexpr int SLAB = 48, RPX = G * SLAB;   // slab rows per window (tokens 0..31, then 32..35 on rows 32, 36, 40, 44) / per workgroup
constexpr int LDX = C + 8;                 // 104 halves
constexpr int XS = RPX * LDX, OS = R * LDX;
constexpr int SMEM96 = (XS + OS) * 2 + R * 8 + 16;
constexpr int LPR = 16, PPR = C / 8, RPP = NTHR / LPR, NPASS = R / RPP;   // row passes: 16 lanes per row (12 carry data), 24 rows per pass, 6 passes
static_assert(R % 16 == 0 && R % RPP == 0 && NPASS == 6, "row passes are reduced as 3 + 3");

__device__ __forceinline__ int slab_row(int t) { return t < 32 ? t : 32 + 4 * (t - 32); }

// P = exp2(S - max) of one query column as fp16 B-operand fragments (keys on the k axis)
__device__ __forceinline__ void probs(const float4v s0, const float4v s1, float s2, float mx, half4& p0, half4& p1, half4& p2) {
    const float2v m2 = {mx, mx};
    const float2v a0 = (float2v){s0[0], s0[1]} - m2, a1 = (float2v){s0[2], s0[3]} - m2;
    const float2v c0 = (float2v){s1[0], s1[1]} - m2, c1 = (float2v){s1[2], s1[3]} - m2;
    p0 = (half4){(_Float16)__builtin_amdgcn_exp2f(a0[0]), (_Float16)__builtin_amdgcn_exp2f(a0[1]), (_Float16)__builtin_amdgcn_exp2f(a1[0]), (_Float16)__builtin_amdgcn_exp2f(a1[1])};
    p1 = (half4){(_Float16)__builtin_amdgcn_exp2f(c0[0]), (_Float16)__builtin_amdgcn_exp2f(c0[1]), (_Float16)__builtin_amdgcn_exp2f(c1[0]), (_Float16)__builtin_amdgcn_exp2f(c1[1])};
    p2 = (half4){(_Float16)__builtin_amdgcn_exp2f(s2 - mx), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};   // slab row 32 + 4g = token 32 + g
}

__global__ __launch_bounds__(NTHR, 4) void swin_attn96_g4_kernel(const SwinAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;              // [RPX][LDX] normalised x slabs; later the output tile [R][LDX] (token order)
    _Float16* Os = Xs + XS;                      // [R][LDX]  attention output, all heads, token order
    int2v* Pix = (int2v*)(Os + OS);              // [R] {source pixel of the token row (-1: none), its slab row}
    int* Cls = (int*)(Pix + R);                  // [G] shift-mask class of each window

    const int tid = threadIdx.x, lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's head
    const int fr = lane & 15, g = lane >> 4;

    const int iw0 = blockIdx.x * G, itotal = p.B * p.nwin;
    const int HW = p.nwin * NTOK;
    const _Float16* __restrict__ X = (const _Float16*)p.x;
    const _Float16* __restrict__ Wqkv = (const _Float16*)p.wqkv_frag;    // [18 row tiles][3 k-steps][64 lanes][8] (engine.cpp frag_major)
    const _Float16* __restrict__ Wproj = (const _Float16*)p.wproj_frag;  // [6 row tiles][3 k-steps][64 lanes][8]
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half8 zero8 = {};

    // weight fragments of head h (q, k, v rows h*16 + fr; 3 k-steps): lane holds [row][ks*32 + 8g .. +7]; stored
    // fragment-major, so each load is one contiguous KiB.  They stay in registers for all four windows.
    half8 wq[3], wk[3], wv[3];
    {
        const _Float16* wlane = Wqkv + lane * 8;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            wq[ks] = *(const half8*)(wlane + (size_t)((0 * NH + h) * 3 + ks) * 512);
            wk[ks] = *(const half8*)(wlane + (size_t)((1 * NH + h) * 3 + ks) * 512);
            wv[ks] = *(const half8*)(wlane + (size_t)((2 * NH + h) * 3 + ks) * 512);
        }
    }

    // ---- source pixel and slab row of every token row, worked out once per workgroup (one thread per row)
    if (tid < R) {
        const int w = tid / NTOK, t = tid - w * NTOK;
        const int iw = iw0 + w;
        int pix = -1;
        if (iw < itotal) {
            const int wb = iw / p.nwin, wl = iw - wb * p.nwin;
            if (p.ry >= 0) {
                const int nwx = p.W / 6;
                const int wy = wl / nwx, wx = wl - wy * nwx;
                const int ty = t / 6, tx = t - ty * 6;
                int y = wy * 6 + ty + p.ry, x = wx * 6 + tx + p.rx;
                y -= y >= p.H ? p.H : 0; x -= x >= p.W ? p.W : 0;
                pix = wb * HW + y * p.W + x;
            } else pix = wb * HW + p.table[wl * NTOK + t];
            if (t == 0) Cls[w] = p.maskid[wl];
        } else if (t == 0) Cls[w] = 0;
        Pix[tid] = (int2v){pix, w * SLAB + slab_row(t)};
    }
    __syncthreads();

    // ---- gather + LayerNorm into the slabs
    {
        const int li = tid & (LPR - 1), rsub = tid / LPR;
        half8 xr[NPASS];
        int srow[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int2v pr = Pix[ps * RPP + rsub];
            srow[ps] = pr[1];
            // rows that do not exist and the four idle lanes of a row read zeros (no masking of the data afterwards)
            const _Float16* src = (pr[0] >= 0 && li < PPR) ? X + (size_t)pr[0] * C + li * 8 : (const _Float16*)kZeroPageG4;
            xr[ps] = *(const half8*)src;
        }
        float sm[NPASS], sq[NPASS];
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
        group_sum16_x6(sm[0], sq[0], sm[1], sq[1], sm[2], sq[2]);
#pragma unroll
        for (int ps = 3; ps < 6; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
        group_sum16_x6(sm[3], sq[3], sm[4], sq[4], sm[5], sq[5]);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const float mean = sm[ps] * (1.f / C);
            const float rstd = rsqrtf(fmaxf(sq[ps] * (1.f / C) - mean * mean, 0.f) + p.eps);
            if (li < PPR) *(half8*)(Xs + srow[ps] * LDX + li * 8) = norm8(xr[ps], rstd, -mean * rstd);
        }
        // the 12 rows between tokens 32..35 of each slab are multiplied like the rest (results ignored): keep them finite
        for (int i = tid; i < G * 12 * PPR; i += NTHR) {
            const int rr = i / PPR, c = i - rr * PPR, w = rr / 12, k = rr - w * 12;
            *(half8*)(Xs + (w * SLAB + 33 + (k / 3) * 4 + (k % 3)) * LDX + c * 8) = zero8;
        }
    }
    __syncthreads();

    const float qscale = p.scale * 1.44269504088896341f;   // log2(e) folded into q: softmax uses exp2
    const float2v qs2 = {qscale, qscale};
    const half4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    const half4 zeroh4 = {};
    // biases of this head: q (initial accumulator of q^T, rows = features 4g..4g+3); the k bias adds the same q.bk to every key
    // of a query and drops out of the softmax; the v bias commutes with the weighted mean (sum p (v + bv) / sum p =
    // sum p v / sum p + bv) and is added to the normalised output instead.
    const float4v bq = *(const float4v*)(p.bqkv + h * HD + g * 4);
    const float4v bv = *(const float4v*)(p.bqkv + 2 * C + h * HD + g * 4);
    const int xoff = fr * LDX + g * 8;                     // this lane's piece of a 16-row operand fragment
    const int wq_ = fr >> 2, qq_ = fr & 3;                 // left-over query tile: column fr = query 32 + qq_ of window wq_

    // ---- left-over queries of the four windows: one q^T tile from a cross-window fragment gather
    half4 qleft;
    float4v sl[3];          // their scores S_left^T [3 key tiles], accumulated over the windows on top of the bias
    float b2l;
    {
        float4v a = bq;
        const _Float16* xq = Xs + (wq_ * SLAB + 32 + 4 * qq_) * LDX + g * 8;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ks], *(const half8*)(xq + ks * 32), a, 0, 0, 0);
        const float2v a0 = (float2v){a[0], a[1]} * qs2, a1 = (float2v){a[2], a[3]} * qs2;
        qleft = (half4){(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
        // bias (+ shift mask) of query tile 2 in the table's lane order (lower.cpp): the entries of query 32 + i sit on lane 16g + i
        const float* lb = p.bias32 + ((size_t)(Cls[wq_] * NH + h) * 3 + 2) * 576;
        const int bl = g * 16 + qq_;
        sl[0] = *(const float4v*)(lb + bl * 4);
        sl[1] = *(const float4v*)(lb + 256 + bl * 4);
        sl[2] = zero4;
        b2l = lb[512 + bl];
    }
    half4 vkeep[G][3];      // v fragments of every window, for the left-over queries' O^T at the end
    half4 pl[3];            // left-over queries' probabilities

#pragma unroll
    for (int w = 0; w < G; ++w) {
        const _Float16* xs = Xs + w * SLAB * LDX + xoff;
        // bias (+ shift mask) of this (window class, head) in load order: per query tile 2 x float4 + 1 float per lane;
        // it is the initial accumulator of S^T
        const int cls = __builtin_amdgcn_readfirstlane(Cls[w]);
        const float* bias = p.bias32 + ((size_t)cls * NH + h) * (3 * 576);
        float4v s[2][3];
        float b2[2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            s[qi][0] = *(const float4v*)(bias + qi * 576 + lane * 4);
            s[qi][1] = *(const float4v*)(bias + qi * 576 + 256 + lane * 4);
            s[qi][2] = zero4;
            b2[qi] = bias[qi * 576 + 512 + lane];         // key tile 2 holds one key per lane: added after the product
        }
        // ---- q^T, k^T (rows = features: A = weights, B = x) and v (rows = slab rows: A = x, B = weights)
        float4v aq[2] = {bq, bq}, ak[3] = {zero4, zero4, zero4}, av[3] = {zero4, zero4, zero4};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const half8 xf = *(const half8*)(xs + tt * 16 * LDX + ks * 32);
                if (tt < 2) aq[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ks], xf, aq[tt], 0, 0, 0);
                ak[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[ks], xf, ak[tt], 0, 0, 0);
                av[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, wv[ks], av[tt], 0, 0, 0);
            }
        }
        half4 qf[2], kf[3];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float2v a0 = (float2v){aq[tt][0], aq[tt][1]} * qs2, a1 = (float2v){aq[tt][2], aq[tt][3]} * qs2;
            qf[tt] = (half4){(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
        }
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { kf[tt][j] = (_Float16)ak[tt][j]; vkeep[w][tt][j] = (_Float16)av[tt][j]; }
        // ---- S^T = K Q^T on top of the bias (k = the 16 features); the left-over queries see this window's keys through the
        // columns of q that belong to it
        const half4 qz = wq_ == w ? qleft : zeroh4;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            s[0][kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qf[0], s[0][kt], 0, 0, 0);
            s[1][kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qf[1], s[1][kt], 0, 0, 0);
            sl[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qz, sl[kt], 0, 0, 0);
        }
        // ---- softmax over the keys (column = query): lane-local maximum of 9, then the 4 lanes of the column
        const float t0 = s[0][2][0] + b2[0], t1 = s[1][2][0] + b2[1];
        float mx0 = max9(s[0][0], s[0][1], t0), mx1 = max9(s[1][0], s[1][1], t1);
        half4 pf[2][3];
        if (w < G - 1) cols_max2(mx0, mx1);
        else {
            const float tl = sl[2][0] + b2l;
            float mxl = max9(sl[0], sl[1], tl);
            cols_max3(mx0, mx1, mxl);
            probs(sl[0], sl[1], tl, mxl, pl[0], pl[1], pl[2]);
        }
        probs(s[0][0], s[0][1], t0, mx0, pf[0][0], pf[0][1], pf[0][2]);
        probs(s[1][0], s[1][1], t1, mx1, pf[1][0], pf[1][1], pf[1][2]);
        // ---- O^T = V^T P^T: rows = features, columns = queries (k = 16 keys per product); parked in Os (token order).
        // The softmax denominators come off the matrix pipe too: a ones matrix in place of V^T leaves the sum of the
        // (fp16) probabilities of query fr in every row of its column - the lane that scales the column already holds it.
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float4v o = __builtin_amdgcn_mfma_f32_16x16x16f16(vkeep[w][0], pf[qi][0], zero4, 0, 0, 0);
            float4v l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf[qi][0], zero4, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x16f16(vkeep[w][1], pf[qi][1], o, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf[qi][1], l, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x16f16(vkeep[w][2], pf[qi][2], o, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf[qi][2], l, 0, 0, 0);
            const float inv = __builtin_amdgcn_rcpf(l[0]);
            const float2v i2 = {inv, inv};
            const float2v o0 = __builtin_elementwise_fma((float2v){o[0], o[1]}, i2, (float2v){bv[0], bv[1]});
            const float2v o1 = __builtin_elementwise_fma((float2v){o[2], o[3]}, i2, (float2v){bv[2], bv[3]});
            *(half4*)(Os + (w * NTOK + qi * 16 + fr) * LDX + h * HD + g * 4) = (half4){(_Float16)o0[0], (_Float16)o0[1], (_Float16)o1[0], (_Float16)o1[1]};
        }
    }
    // ---- left-over queries: O^T against every window's V, each column keeps the product with its own window
    {
        float4v l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pl[0], zero4, 0, 0, 0);
        l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pl[1], l, 0, 0, 0);
        l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pl[2], l, 0, 0, 0);
        float4v o = zero4;
#pragma unroll
        for (int w = 0; w < G; ++w) {
            float4v ow = __builtin_amdgcn_mfma_f32_16x16x16f16(vkeep[w][0], pl[0], zero4, 0, 0, 0);
            ow = __builtin_amdgcn_mfma_f32_16x16x16f16(vkeep[w][1], pl[1], ow, 0, 0, 0);
            ow = __builtin_amdgcn_mfma_f32_16x16x16f16(vkeep[w][2], pl[2], ow, 0, 0, 0);
            o = wq_ == w ? ow : o;
        }
        const float inv = __builtin_amdgcn_rcpf(l[0]);
        const float2v i2 = {inv, inv};
        const float2v o0 = __builtin_elementwise_fma((float2v){o[0], o[1]}, i2, (float2v){bv[0], bv[1]});
        const float2v o1 = __builtin_elementwise_fma((float2v){o[2], o[3]}, i2, (float2v){bv[2], bv[3]});
        *(half4*)(Os + (wq_ * NTOK + 32 + qq_) * LDX + h * HD + g * 4) = (half4){(_Float16)o0[0], (_Float16)o0[1], (_Float16)o1[0], (_Float16)o1[1]};
    }
    __syncthreads();      // every head's outputs are in Os; nobody reads the slabs any more

    // the residual rows are fetched now, under the projection
    half8 xres[NPASS];
    int my_pix[NPASS];
    {
        const int li = tid & (LPR - 1), rsub = tid / LPR;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            my_pix[ps] = Pix[ps * RPP + rsub][0];
            const _Float16* src = (my_pix[ps] >= 0 && li < PPR) ? X + (size_t)my_pix[ps] * C + li * 8 : (const _Float16*)kZeroPageG4;
            xres[ps] = *(const half8*)src;
        }
    }
    // ---- proj, transposed: out^T = Wproj Os^T + b (rows = output channels, columns = tokens), so a lane ends with 4 consecutive
    // channels of one token.  18 units of (16-token tile, 3 channel tiles), 3 per wave; weights as fragments from L2, bias as the
    // initial accumulator; the tile goes over Xs in token order.
#pragma unroll 1
    for (int u = h; u < RT * 2; u += NH) {
        const int mt = u >> 1, n3 = (u & 1) * 3;
        half8 wf[3][3];
        float4v acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wf[t][ks] = *(const half8*)(Wproj + (size_t)(((n3 + t) * 3 + ks) * 64 + lane) * 8);
            acc[t] = *(const float4v*)(p.bproj + (n3 + t) * 16 + g * 4);
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const half8 of = *(const half8*)(Os + mt * 16 * LDX + xoff + ks * 32);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][ks], of, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
            *(half4*)(Xs + (mt * 16 + fr) * LDX + (n3 + t) * 16 + g * 4) = (half4){(_Float16)acc[t][0], (_Float16)acc[t][1], (_Float16)acc[t][2], (_Float16)acc[t][3]};
    }
    __syncthreads();

    // ---- row pieces: + residual x, scatter store, LayerNorm statistics for the next op
    {
        _Float16* __restrict__ Y = (_Float16*)p.y;
        const int li = tid & (LPR - 1), rsub = tid / LPR;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + rsub;
            const int pix = my_pix[ps];
            const bool ok = pix >= 0 && li < PPR;
            half8 o = {};
            if (ok) {
                o = *(const half8*)(Xs + r * LDX + li * 8) + xres[ps];
                *(half8*)(Y + (size_t)pix * C + li * 8) = o;
            }
            if (p.stats_out) {
                float s, q;
                sum_sq8(o, s, q);
                s = group_sum16(s);
                q = group_sum16(q);
                const float mean = s * (1.f / C);
                if (ok && li == 0) { p.stats_out[2 * (size_t)pix] = mean; p.stats_out[2 * (size_t)pix + 1] = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out); }
            }
        }
    }
}

}  // namespace

hipError_t launch_swin_attn96_g4(const SwinAttnParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)swin_attn96_g4_kernel, SMEM96, lds_ok); e != hipSuccess) return e;
    const long total_win = (long)p.B * p.nwin;
    dim3 grid((unsigned)((total_win + G - 1) / G));
    hipLaunchKernelGGL(swin_attn96_g4_kernel, grid, dim3(NTHR), SMEM96, s, p);
    return hipGetLastError();
}

}  // namespace w2x
