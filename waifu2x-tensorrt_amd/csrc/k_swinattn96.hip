// Fused Swin attention branch, C = 96 / 6 heads of 16 / window 6x6 for gfx950:  y = x + proj( W-MSA( LayerNorm(x) ) )
// (LayerNormalization, roll + window partition, QKV MatMul+Add, per-head scale / QK^T / rel-pos bias (+ shift mask) / Softmax /
//  .V, head merge, proj MatMul+Add, window reverse + reverse roll, residual Add of the ONNX graph: one launch.)
//
// A workgroup = 4 waves = 2 windows (72 token rows), four workgroups per CU (one wave of each per SIMD: workgroups whose wave
// count is not a multiple of four load the SIMDs unevenly - a 6-wave variant measured that in round 2: git history, tools/ab/k_swinattn96_g4.hip).  The 12 (window, head)
// units of a workgroup go 3 to a wave so that two of them share a head:
//     wave v:  unit 0 = (window 0, head v),  unit 1 = (window 1, head v),  unit 2 = (window v & 1, head 4 + (v >> 1)).
// The q/k/v weight fragments of a head are read once per wave from the fragment-major copy (one contiguous KiB per load) and
// serve both windows; the second head's fragments take over the same registers after unit 1.  q, k, v, S and P never leave
// registers: with a head dimension of 16 the accumulator layout (lane (col, g) holds rows 4g..4g+3) is the operand layout of
// v_mfma_f32_16x16x16_f16, so
//     q^T, k^T (rows = features)  -> B / A operands of S^T = K Q^T,
//     v (rows = tokens)           -> A operand of O^T = V^T P^T,
//     S^T after the softmax       -> B operand of O^T.
// A window's 36 tokens are two full 16-token tiles plus 4 left over.  Left-over KEYS sit on rows 0,4,8,12 of a third key tile
// (slab rows 32,36,40,44), so a lane holds exactly one of them.  Left-over QUERIES of a wave's three units share ONE query tile
// (column 4u+i = query 32+i of unit u): their q comes from a cross-window fragment gather (one product per head), their scores
// are accumulated over the units with the other units' columns of q zeroed (S_left += K_u Qz_u), and they take a single softmax
// pass next to the last unit's instead of three mostly empty ones.
// The kernel is instruction-issue bound (SQ counters: three waves per SIMD, each ~31 % issuing), so everything else is about
// instructions per token: rows that do not exist are loaded from a zero page instead of being masked, LayerNorm scaling on
// v_fma_mix (f16 in, f32 math, f16 out: one instruction per element), the row -> (pixel, slab row) map computed once per
// workgroup, softmax denominators from a ones-operand MFMA, q bias as the initial accumulator, k bias dropped / v bias after
// the normalisation, proj computed transposed so that a lane ends with 4 consecutive channels of one token (bias as the
// initial accumulator, 8-byte LDS stores).
#include "kernels.h"

#include <algorithm>

// Choices that were build switches while they were being measured (round 3 - 5; the variants are in git history at 9576837, the records under
// profiles/r5_kernels/a96_*.txt, profiles/r3_kernels/attn96_*.txt):
//   * O^T = V^T P^T over the 32 keys of the two full key tiles as ONE v_mfma_f32_16x16x32_f16 (the accumulator tiles of S^T and v pair up into its
//     operands with the k order permuted identically on both sides) + one 16x16x16 for the left-over keys, instead of three 16x16x16 - which cost the
//     matrix pipe as much as a 16x16x32 each (tools/issue_model.hip): 9 matrix instructions less per wave (0.510 -> 0.496 ms per launch).  The softmax
//     denominators the same way measured level (0.498) and keep their three 16x16x16.
//     The two shapes NEVER share an accumulator: a 16x16x16 whose SrcC is the vDst of the 16x16x32 right in front of it gets no wait states from hipcc
//     (it treats the pair like two instructions of one opcode, which the hardware forwards), and the second product then reads a half-written
//     accumulator - outputs off by up to 0.8 and different from run to run in three of the four builds that had such a pair
//     (profiles/r5_kernels/a96_mixed_chain.txt; tools/isa_mfma_chain.py finds the pairs in the ISA, tests/test_isa_hazards.py keeps them out of the
//     library).  Each shape accumulates into its own registers and the two sums meet in the epilogue's fused multiply-adds.
//   * weight fragments, bias tables and the rel-pos bias through buffer loads (lane offset in a VGPR once, everything else in the scalar offset)
//     instead of flat loads with 64-bit per-lane addresses.
//   * the q / k / v bias vectors are copied to LDS when the workgroup starts and read from there (with the 16x16x32 PV the allocation fits 126
//     registers and the bias loads leave the vector-memory queue).
//   * the residual rows are requested after the last unit's q / k / v products (in front of the projection: 0.524 ms in round 3; after the unit's score
//     products 0.502 against 0.496).
//   * s_setprio by phase (head loop at 1, or rising with progress): +15 % - off.

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// Rows are fetched and stored through buffer resources over x / y: an offset at or beyond num_records reads zeros and drops
// stores, so rows that do not exist (and the four idle lanes of a 16-lane row) need neither a predicate nor masking of the data.
// Offsets are 32 bits: the launcher cuts passes of more than kMaxBufBytes into runs of whole images.
constexpr unsigned kNoRow = 0xFFFFFFFFu;     // saturating adds keep it there
constexpr size_t kMaxBufBytes = 0xFFFFFF00u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);   // raw buffer, 32-bit offsets, bounds-checked
}

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
// Row-group sums of four / six independent values at once: v_add_f32 with a DPP operand; the chains are interleaved so that each
// one's two wait states between a VALU write and a DPP read are filled by the others.
#define W2X_DPP1(R, CTRL) "v_add_f32_dpp " R ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define W2X_DPP6(CTRL) W2X_DPP1("%0", CTRL) W2X_DPP1("%1", CTRL) W2X_DPP1("%2", CTRL) W2X_DPP1("%3", CTRL) W2X_DPP1("%4", CTRL) W2X_DPP1("%5", CTRL)
#define W2X_DPP4(CTRL) W2X_DPP1("%0", CTRL) W2X_DPP1("%1", CTRL) W2X_DPP1("%2", CTRL) W2X_DPP1("%3", CTRL)
__device__ __forceinline__ void group_sum16_x4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 2\n\t" W2X_DPP4("quad_perm:[1,0,3,2]") W2X_DPP4("quad_perm:[2,3,0,1]") W2X_DPP4("row_half_mirror") W2X_DPP4("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void group_sum16_x6(float& a, float& b, float& c, float& d, float& e, float& f) {
    asm volatile("s_nop 2\n\t" W2X_DPP6("quad_perm:[1,0,3,2]") W2X_DPP6("quad_perm:[2,3,0,1]") W2X_DPP6("row_half_mirror") W2X_DPP6("row_mirror")
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

constexpr int C = 96, HD = 16, NH = 6, NTOK = 36, G = 2, R = G * NTOK, RT = 5, RP = RT * 16;   // 72 rows in 5 row tiles
constexpr int NTHR = 256, NU = 3;          // 4 waves, 3 (window, head) units per wave
constexpr int SLAB = 48, RPX = G * SLAB;   // slab rows per window (tokens 0..31, then 32..35 on rows 32, 36, 40, 44) / per workgroup
constexpr int LDX = C + 8;                 // 104 halves
constexpr int XS = RPX * LDX, OS = RP * LDX;
constexpr int NPAD = G * 12;                // slab rows between the left-over tokens (kept at zero)
constexpr int BQ_OFF = (XS + OS) * 2 + (RP + NPAD) * 8 + 16;   // q / k / v bias [3 * C] fp32, copied in when the workgroup starts
constexpr int SMEM96 = BQ_OFF + 3 * C * 4;
constexpr int DUMMY = XS * 2;              // byte offset of a row nobody reads at that point (first row of Os): target of the stores of lanes / rows without data
constexpr int LPR = 16, PPR = C / 8, RPP = NTHR / LPR, NPASS = RP / RPP;   // row passes: 16 lanes per row (12 carry data), 16 rows per pass, 5 passes
static_assert(NPASS == 5, "the row sums are reduced as 3 + 2 passes");

__device__ __forceinline__ int slab_row(int t) { return t < 32 ? t : 32 + 4 * (t - 32); }

// P = exp2(S - max) of one query column as fp16 B-operand fragments (keys on the k axis): p01 = key tiles 0 and 1 (k slot j < 4: key
// 4g + j, j >= 4: key 16 + 4g + j - 4 - the order concat(v tile 0, v tile 1) has too), p2 = the left-over keys
__device__ __forceinline__ void probs(const float4v s0, const float4v s1, float s2, float mx, half8& p01, half4& p2) {
    const float2v m2 = {mx, mx};
    const float2v a0 = (float2v){s0[0], s0[1]} - m2, a1 = (float2v){s0[2], s0[3]} - m2;
    const float2v c0 = (float2v){s1[0], s1[1]} - m2, c1 = (float2v){s1[2], s1[3]} - m2;
    p01 = (half8){(_Float16)__builtin_amdgcn_exp2f(a0[0]), (_Float16)__builtin_amdgcn_exp2f(a0[1]), (_Float16)__builtin_amdgcn_exp2f(a1[0]), (_Float16)__builtin_amdgcn_exp2f(a1[1]),
                  (_Float16)__builtin_amdgcn_exp2f(c0[0]), (_Float16)__builtin_amdgcn_exp2f(c0[1]), (_Float16)__builtin_amdgcn_exp2f(c1[0]), (_Float16)__builtin_amdgcn_exp2f(c1[1])};
    p2 = (half4){(_Float16)__builtin_amdgcn_exp2f(s2 - mx), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};   // slab row 32 + 4g = token 32 + g
}
__device__ __forceinline__ half4 lo4(const half8 v) { return (half4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ half4 hi4(const half8 v) { return (half4){v[4], v[5], v[6], v[7]}; }
// product over the 36 keys: operands (a01 | a2) x (p01 | p2), added to `acc`: .a = the 32 keys of the full tiles (16x16x32), .b = the left-over keys
// (16x16x16) - one accumulator per instruction shape (header)
struct KeysAcc { float4v a, b; };
__device__ __forceinline__ KeysAcc keys_product(const half8 a01, const half4 a2, const half8 p01, const half4 p2, const KeysAcc acc) {
    KeysAcc o;
    o.a = __builtin_amdgcn_mfma_f32_16x16x32_f16(a01, p01, acc.a, 0, 0, 0);
    o.b = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, p2, acc.b, 0, 0, 0);
    return o;
}
// (o.a + o.b) * inv + bv as the four fp16 values a lane stores
__device__ __forceinline__ half4 scaled_output(const KeysAcc o, float inv, const float4v bv) {
    const float2v i2 = {inv, inv};
    float2v o0 = __builtin_elementwise_fma((float2v){o.a[0], o.a[1]}, i2, (float2v){bv[0], bv[1]});
    float2v o1 = __builtin_elementwise_fma((float2v){o.a[2], o.a[3]}, i2, (float2v){bv[2], bv[3]});
    o0 = __builtin_elementwise_fma((float2v){o.b[0], o.b[1]}, i2, o0);
    o1 = __builtin_elementwise_fma((float2v){o.b[2], o.b[3]}, i2, o1);
    return (half4){(_Float16)o0[0], (_Float16)o0[1], (_Float16)o1[0], (_Float16)o1[1]};
}
// column sums of P (the softmax denominators) off the matrix pipe: a ones matrix in place of V^T; three 16x16x16 products share a two-register ones operand
__device__ __forceinline__ float keys_sum(const half8 p01, const half4 p2) {
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    float4v l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, lo4(p01), zero4, 0, 0, 0);
    l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, hi4(p01), l, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(ones, p2, l, 0, 0, 0)[0];
}

// four waves per SIMD (= four workgroups per CU): at three (up to 168 registers) every combination of the choices in the header measured 4.5 - 10 % slower
// (profiles/r5_kernels/a96_*.txt, a96_occupancy.txt) - the kernel lives on its fourth wave per SIMD
__global__ __launch_bounds__(NTHR, 4) void swin_attn96_kernel(const SwinAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;              // [RPX][LDX] normalised x slabs; later the output tile [RP][LDX] (token order)
    _Float16* Os = Xs + XS;                      // [RP][LDX]  attention output, all heads, token order
    int2v* Pix = (int2v*)(Os + OS);              // [RP] {byte offset of the token row's pixel in x / y (kNoRow: none), byte offset of its slab row}, then [NPAD] {-, offset of a pad row}
    int* Cls = (int*)(Pix + RP + NPAD);          // [G] shift-mask class of each window

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const int wl0 = blockIdx.x * G;              // the workgroup's windows: wl0 and wl0 + 1 of image blockIdx.y (a pair never straddles two images: no division by the window count)
    const int HW = p.nwin * NTOK;
    const unsigned xbytes = (unsigned)p.B * (unsigned)HW * (C * 2);
    const __amdgpu_buffer_rsrc_t X = make_rsrc(p.x, xbytes), Y = make_rsrc(p.y, xbytes);
    const _Float16* __restrict__ Wqkv = (const _Float16*)p.wqkv_frag;    // [18 row tiles][3 k-steps][64 lanes][8] (engine.cpp frag_major)
    const _Float16* __restrict__ Wproj = (const _Float16*)p.wproj_frag;  // [6 row tiles][3 k-steps][64 lanes][8]
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half8 zero8 = {};

    // this wave's units: heads hA (both windows) and hC (window wC)
    const int hA = wv, hC = 4 + (wv >> 1), wC = wv & 1;

    // weight fragments of a head (q, k, v rows h*16 + fr; 3 k-steps): lane holds [row][ks*32 + 8g .. +7]; stored
    // fragment-major, so each load is one contiguous KiB
    half8 wq[3], wk[3], wv_[3];
    // every table goes through a buffer resource: the lane part of the address is one VGPR for all loads, the rest is scalar
    const __amdgpu_buffer_rsrc_t WQ = make_rsrc(Wqkv, 3u * C * C * 2u), WP = make_rsrc(Wproj, (unsigned)C * C * 2u);
    const __amdgpu_buffer_rsrc_t BP = make_rsrc(p.bproj, (unsigned)C * 4u), RB = make_rsrc(p.bias32, 0x7FFFFFF0u);
    const unsigned l16 = lane * 16u, l4 = lane * 4u, g16 = (lane >> 4) * 16u;
#define W2X_WFRAG(SEL, H, KS) __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WQ, l16, (unsigned)(((SEL) * NH + (H)) * 3 + (KS)) * 1024u, 0))
#define W2X_BQKV(OFF) (*(const float4v*)((const float*)(smem + BQ_OFF) + (OFF) + g * 4))
    if (tid < 3 * C / 4) *(float4v*)(smem + BQ_OFF + tid * 16) = *(const float4v*)(p.bqkv + tid * 4);   // (first use is two barriers away)
#define W2X_LOAD_W(H)                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 3; ++ks) {                                          \
        wq[ks] = W2X_WFRAG(0, H, ks);                                                           \
        wk[ks] = W2X_WFRAG(1, H, ks);                                                           \
        wv_[ks] = W2X_WFRAG(2, H, ks);                                                          \
    }
    W2X_LOAD_W(hA)

    // ---- source pixel and slab row of every token row, worked out once per workgroup (one thread per row)
    if (tid < RP + NPAD) {
        int pix = -1, srow = DUMMY;                      // pix: pixel index, stored as a byte offset
        if (tid >= RP) {            // the 12 rows between tokens 32..35 of each slab
            const int k = tid - RP, w = k >= 12 ? 1 : 0, kk = k - 12 * w;
            srow = (w * SLAB + 33 + (kk / 3) * 4 + (kk % 3)) * LDX * 2;
        } else if (tid < R) {
            const int w = tid >= NTOK ? 1 : 0, t = tid - w * NTOK;
            const int wl = wl0 + w;
            srow = (w * SLAB + slab_row(t)) * LDX * 2;
            if (wl < p.nwin) {
                if (p.ry >= 0) {
                    // window -> (wy, wx) through a reciprocal instead of an integer division (two of which were a third of this block's instructions):
                    // (wl + 0.5) / nwx is at least 0.5 / nwx away from an integer, far more than the rounding of the product for any token map a pass holds
                    const int nwx = p.W / 6;
                    const int wy = (int)(((float)wl + 0.5f) * __builtin_amdgcn_rcpf((float)nwx)), wx = wl - wy * nwx;
                    const int ty = (t * 43) >> 8, tx = t - ty * 6;           // t / 6 for t < 36
                    int y = wy * 6 + ty + p.ry, x = wx * 6 + tx + p.rx;
                    y -= y >= p.H ? p.H : 0; x -= x >= p.W ? p.W : 0;
                    pix = (int)blockIdx.y * HW + y * p.W + x;
                } else pix = (int)blockIdx.y * HW + p.table[wl * NTOK + t];
                if (t == 0) Cls[w] = p.maskid[wl];
            } else if (t == 0) Cls[w] = 0;
        }
        Pix[tid] = (int2v){pix < 0 ? (int)kNoRow : (int)((unsigned)pix * (unsigned)(C * 2)), srow};   // offsets are unsigned 32-bit (up to 4 GB per run)
    }
    __syncthreads();

    // ---- gather + LayerNorm into the slabs
    {
        const int li = tid & (LPR - 1), rsub = tid / LPR;
        const unsigned lane_off = li < PPR ? li * 16u : kNoRow;
        half8 xr[NPASS];
        int srow[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int2v pr = Pix[ps * RPP + rsub];
            srow[ps] = pr[1];
            // rows that do not exist and the four idle lanes of a row read zeros
            xr[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, __builtin_elementwise_add_sat((unsigned)pr[0], lane_off), 0, 0));
        }
        float sm[NPASS], sq[NPASS];
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
        group_sum16_x6(sm[0], sq[0], sm[1], sq[1], sm[2], sq[2]);
#pragma unroll
        for (int ps = 3; ps < 5; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
        group_sum16_x4(sm[3], sq[3], sm[4], sq[4]);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const float mean = sm[ps] * (1.f / C);
            const float rstd = __builtin_amdgcn_rsqf(fmaxf(sq[ps] * (1.f / C) - mean * mean, 0.f) + p.eps);   // the argument is >= eps: no denormal scaling needed
            // unconditional store: the four idle lanes of a row and rows without a token write to the dummy row
            *(half8*)(smem + (li < PPR ? srow[ps] + li * 16 : DUMMY)) = norm8(xr[ps], rstd, -mean * rstd);
        }
        // the 12 rows between tokens 32..35 of each slab are multiplied like the rest (results ignored): keep them finite
#pragma unroll
        for (int k = 0; k < (NPAD + RPP - 1) / RPP; ++k) {
            const int pr = k * RPP + rsub;
            if (pr < NPAD) *(half8*)(smem + (li < PPR ? Pix[RP + pr][1] + li * 16 : DUMMY)) = zero8;
        }
    }
    __syncthreads();

    const float qscale = p.scale * 1.44269504088896341f;   // log2(e) folded into q: softmax uses exp2
    const float2v qs2 = {qscale, qscale};
    const half4 zeroh4 = {};
    const int xoff = fr * LDX + g * 8;                     // this lane's piece of a 16-row operand fragment
    // left-over query tile: column fr = query 32 + ql of unit ul (ul = 3: unused column, computed like unit 0's and never stored)
    const int ul = fr >> 2, ql = fr & 3;
    const int wl_ = ul == 1 ? 1 : ul == 2 ? wC : 0;        // its window
    const int hl_ = ul == 2 ? hC : hA;                     // its head
    const _Float16* xq = Xs + (wl_ * SLAB + 32 + 4 * ql) * LDX + g * 8;

    half4 qleft;            // q of the left-over queries (columns of units 0, 1 now; unit 2's are merged in when its weights are there)
    float4v sl[3];          // their scores S_left^T [3 key tiles], accumulated over the units on top of the bias
    float b2l;
    {
        float4v a = W2X_BQKV(hA * HD);   // q bias = initial accumulator
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ks], *(const half8*)(xq + ks * 32), a, 0, 0, 0);
        const float2v a0 = (float2v){a[0], a[1]} * qs2, a1 = (float2v){a[2], a[3]} * qs2;
        qleft = (half4){(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
        // bias (+ shift mask) of query tile 2 in the table's lane order (lower.cpp): the entries of query 32 + i sit on lane 16g + i
        const float* lb = p.bias32 + ((size_t)(Cls[wl_] * NH + hl_) * 3 + 2) * 576;
        const int bl = g * 16 + ql;
        sl[0] = *(const float4v*)(lb + bl * 4);
        sl[1] = *(const float4v*)(lb + 256 + bl * 4);
        sl[2] = zero4;
        b2l = lb[512 + bl];
    }
    // The residual rows (the same pixels again, now for y = x + ...): requested when the last unit's q / k / v products are done - its
    // weight registers are free from there on - so that the fetch travels under that unit's softmax and the left-over queries instead
    // of in front of the projection's weight loads (loads return in order: the first projection product would wait for these rows).
    half8 xres[NPASS];
#define W2X_FETCH_XRES() {                                                                                          \
        const int li_ = tid & (LPR - 1), rsub_ = tid / LPR;                                                         \
        const unsigned lane_off_ = li_ < PPR ? li_ * 16u : kNoRow;                                                  \
        _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps)                                                        \
            xres[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, __builtin_elementwise_add_sat((unsigned)Pix[ps * RPP + rsub_][0], lane_off_), 0, W2X_LD_LAST_AUX)); }
    half8 vk01[NU]; half4 vk2[NU];   // v fragments of every unit (key tiles 0 | 1, left-over keys), also for the left-over queries' O^T at the end
    half8 pl01; half4 pl2;           // left-over queries' probabilities

#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int w = u == 2 ? wC : u, h = u == 2 ? hC : hA;   // (window, head) of this unit
        const _Float16* xs = Xs + w * SLAB * LDX + xoff;
        // biases of this head: q (initial accumulator of q^T, rows = features 4g..4g+3); the k bias adds the same q.bk to every
        // key of a query and drops out of the softmax; the v bias commutes with the weighted mean (sum p (v + bv) / sum p =
        // sum p v / sum p + bv) and is added to the normalised output instead.
        const float4v bq = W2X_BQKV(h * HD);
        const float4v bv = W2X_BQKV(2 * C + h * HD);
        if (u == 2) {   // unit 2's left-over queries: q with the second head's weights, merged into its columns
            float4v a = bq;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ks], *(const half8*)(xq + ks * 32), a, 0, 0, 0);
            const float2v a0 = (float2v){a[0], a[1]} * qs2, a1 = (float2v){a[2], a[3]} * qs2;
            const half4 qc = {(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
            qleft = ul == 2 ? qc : qleft;
        }
        // bias (+ shift mask) of this (window class, head) in load order: per query tile 2 x float4 + 1 float per lane;
        // it is the initial accumulator of S^T
        const int cls = __builtin_amdgcn_readfirstlane(Cls[w]);
        float4v s[2][3];
        float b2[2];
        const unsigned boff = (unsigned)(cls * NH + h) * (3u * 576u * 4u);
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            s[qi][0] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(RB, l16, boff + (unsigned)(qi * 576) * 4u, 0));
            s[qi][1] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(RB, l16, boff + (unsigned)(qi * 576 + 256) * 4u, 0));
            s[qi][2] = zero4;
            b2[qi] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(RB, l4, boff + (unsigned)(qi * 576 + 512) * 4u, 0));   // key tile 2 holds one key per lane: added after the product
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
                av[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, wv_[ks], av[tt], 0, 0, 0);
            }
        }
        if (u == 1) { W2X_LOAD_W(hC) }   // the second head's fragments take over the registers (reloading each register right after its
                                         // last use inside the loop above measured no faster and cost a spill)
        if (u == NU - 1) W2X_FETCH_XRES()
        half4 qf[2], kf[3];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float2v a0 = (float2v){aq[tt][0], aq[tt][1]} * qs2, a1 = (float2v){aq[tt][2], aq[tt][3]} * qs2;
            qf[tt] = (half4){(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
        }
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) kf[tt][j] = (_Float16)ak[tt][j];
        vk01[u] = (half8){(_Float16)av[0][0], (_Float16)av[0][1], (_Float16)av[0][2], (_Float16)av[0][3], (_Float16)av[1][0], (_Float16)av[1][1], (_Float16)av[1][2], (_Float16)av[1][3]};
        vk2[u] = (half4){(_Float16)av[2][0], (_Float16)av[2][1], (_Float16)av[2][2], (_Float16)av[2][3]};
        // ---- S^T = K Q^T on top of the bias (k = the 16 features); the left-over queries see this unit's keys through the
        // columns of q that belong to it
        const half4 qz = ul == u ? qleft : zeroh4;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            s[0][kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qf[0], s[0][kt], 0, 0, 0);
            s[1][kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qf[1], s[1][kt], 0, 0, 0);
            sl[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qz, sl[kt], 0, 0, 0);
        }
        // ---- softmax over the keys (column = query): lane-local maximum of 9, then the 4 lanes of the column
        const float t0 = s[0][2][0] + b2[0], t1 = s[1][2][0] + b2[1];
        float mx0 = max9(s[0][0], s[0][1], t0), mx1 = max9(s[1][0], s[1][1], t1);
        half8 pf01[2]; half4 pf2[2];
        if (u < NU - 1) cols_max2(mx0, mx1);
        else {
            const float tl = sl[2][0] + b2l;
            float mxl = max9(sl[0], sl[1], tl);
            cols_max3(mx0, mx1, mxl);
            probs(sl[0], sl[1], tl, mxl, pl01, pl2);
        }
        probs(s[0][0], s[0][1], t0, mx0, pf01[0], pf2[0]);
        probs(s[1][0], s[1][1], t1, mx1, pf01[1], pf2[1]);
        // ---- O^T = V^T P^T: rows = features, columns = queries; parked in Os (token order).
        // The softmax denominators come off the matrix pipe too: a ones matrix in place of V^T leaves the sum of the
        // (fp16) probabilities of query fr in every row of its column - the lane that scales the column already holds it.
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const KeysAcc o = keys_product(vk01[u], vk2[u], pf01[qi], pf2[qi], KeysAcc{zero4, zero4});
            const float inv = __builtin_amdgcn_rcpf(keys_sum(pf01[qi], pf2[qi]));
            *(half4*)(Os + (w * NTOK + qi * 16 + fr) * LDX + h * HD + g * 4) = scaled_output(o, inv, bv);
        }
    }
#undef W2X_LOAD_W
#undef W2X_WFRAG
#undef W2X_BQKV
    // ---- left-over queries: O^T against every unit's V; a column only carries probabilities into the product with its own unit, so the three
    // products accumulate into one output tile
    {
        const float inv = __builtin_amdgcn_rcpf(keys_sum(pl01, pl2));
        KeysAcc o = {zero4, zero4};
#pragma unroll
        for (int u = 0; u < NU; ++u) o = keys_product(vk01[u], vk2[u], ul == u ? pl01 : zero8, ul == u ? pl2 : zeroh4, o);
        const float4v bv = *(const float4v*)(p.bqkv + 2 * C + hl_ * HD + g * 4);   // (hl_ differs per lane)
        if (ul < NU) *(half4*)(Os + (wl_ * NTOK + 32 + ql) * LDX + hl_ * HD + g * 4) = scaled_output(o, inv, bv);
    }
    __syncthreads();      // every head's outputs are in Os; nobody reads the slabs any more

#undef W2X_FETCH_XRES
    // ---- proj, transposed: out^T = Wproj Os^T + b (rows = output channels, columns = tokens), so a lane ends with 4 consecutive
    // channels of one token.  10 units of (16-token tile, 3 channel tiles); weights as fragments from L2, bias as the initial
    // accumulator; the tile goes over Xs in token order.
#pragma unroll 1
    for (int u = wv; u < RT * 2; u += 4) {
        const int mt = u >> 1, n3 = (u & 1) * 3;
        half8 wf[3][3];
        float4v acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wf[t][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WP, l16, (unsigned)((n3 + t) * 3 + ks) * 1024u, 0));
            acc[t] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(BP, g16, (unsigned)(n3 + t) * 64u, 0));
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
        const int li = tid & (LPR - 1), rsub = tid / LPR;
        const unsigned lane_off = li < PPR ? li * 16u : kNoRow;
        unsigned my_off[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) my_off[ps] = __builtin_elementwise_add_sat((unsigned)Pix[ps * RPP + rsub][0], lane_off);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + rsub;
            const bool ok = my_off[ps] != kNoRow;
            // (idle lanes read the next row's first pieces - any finite or non-finite bits: their store is dropped)
            half8 o = *(const half8*)(Xs + r * LDX + li * 8) + xres[ps];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), Y, my_off[ps], 0, W2X_ST_AUX);
            if (p.stats_out) {
                if (!ok) o = half8{};
                const size_t pix = my_off[ps] / (C * 2);
                float s, q;
                sum_sq8(o, s, q);
                s = group_sum16(s);
                q = group_sum16(q);
                const float mean = s * (1.f / C);
                if (ok && li == 0) { p.stats_out[2 * (size_t)pix] = mean; p.stats_out[2 * (size_t)pix + 1] = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out); }
            }
        }
    }
}

}  // namespace

hipError_t launch_swin_attn96(const SwinAttnParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)swin_attn96_kernel, SMEM96, lds_ok); e != hipSuccess) return e;
    // the kernel addresses x / y with 32-bit byte offsets: passes beyond that are cut into runs of whole images (windows never
    // cross an image, the statistics rows follow the pixels)
    const size_t img_bytes = (size_t)p.nwin * NTOK * C * 2;
    if (img_bytes == 0 || img_bytes > kMaxBufBytes) return hipErrorInvalidValue;
    const int per_run = (int)std::min<size_t>((size_t)p.B, kMaxBufBytes / img_bytes);
    for (int b0 = 0; b0 < p.B; b0 += per_run) {
        SwinAttnParams q = p;
        q.B = std::min(per_run, p.B - b0);
        q.x = (const char*)p.x + (size_t)b0 * img_bytes;
        q.y = (char*)p.y + (size_t)b0 * img_bytes;
        if (p.stats_out) q.stats_out = p.stats_out + (size_t)b0 * p.nwin * NTOK * 2;
        hipLaunchKernelGGL(swin_attn96_kernel, dim3((unsigned)((q.nwin + G - 1) / G), (unsigned)q.B), dim3(NTHR), SMEM96, s, q);      // x: window pairs of an image, y: images
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace w2x
