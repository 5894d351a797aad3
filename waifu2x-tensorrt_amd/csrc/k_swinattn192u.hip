// Fused Swin attention branch, C = 192 / 6 heads of 32 / window 6x6 - the THREE-WORKGROUPS-PER-CU variant of git 9576837:tools/ab/k_swinattn192_r3.hip (round 4).
//     y = x + proj( W-MSA( LayerNorm(x) ) )
// Same math, same instruction-level recipe per (window, head) unit and the same bytes as git 9576837:tools/ab/k_swinattn192_r3.hip; what changes is what a wave
// keeps alive.  The round-2/3 kernel lets wave v multiply head v's weights against BOTH windows at once (every weight fragment feeds six
// products) and parks the head outputs in a second LDS tile: 230+ registers and 70 KB of LDS per workgroup = two workgroups (two waves per
// SIMD) per CU, where the kernel waits more than it issues (SQ_WAIT_INST_ANY + SQ_WAIT_ANY = 63 % of its wave cycles, profiles/r3_late).
// Here
//   * a wave works through its three units one at a time (a unit = q / k / v products of one head on ONE window, then its attention): the
//     second window's products of a head stream that head's 36 weight fragments from L2 again instead of holding both windows' q, k, v;
//   * the head outputs stay in registers (12 per unit) until every wave has finished reading the slabs, and then go where the slabs were:
//     LDS per workgroup is the slab area plus tables, 41.5 KB;
//   * the projection keeps its 15 output tiles in registers across the barrier that frees the head-output tile, which it then overwrites.
// That is <= 168 registers and three workgroups per CU (three waves per SIMD), for two more workgroup barriers and a third more weight
// traffic from L2.  tools/ab/attn192_variants.sh compares it with the shipped kernel bit for bit and times both.
#include "kernels.h"

#include <algorithm>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

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
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
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
// Rows are fetched and stored through buffer resources over x / y (k_swinattn96.hip): an offset at or beyond num_records reads zeros
// and drops stores, so rows that do not exist and the idle lanes of a row need neither a predicate nor masking of the data.
constexpr unsigned kNoRow = 0xFFFFFFFFu;     // saturating adds keep it there
constexpr size_t kMaxBufBytes = 0xFFFFFF00u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// see k_swinattn.hip for why the swaps are inline asm on two registers
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
// 32-lane group sums of six independent values at once (k_swinattn96.hip): four v_add_f32 steps with a DPP operand
// inside the 16-lane rows, then one row swap across; the chains fill each other's wait states.
#define W2X_DPP1(R, CTRL) "v_add_f32_dpp " R ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define W2X_DPP6(CTRL) W2X_DPP1("%0", CTRL) W2X_DPP1("%1", CTRL) W2X_DPP1("%2", CTRL) W2X_DPP1("%3", CTRL) W2X_DPP1("%4", CTRL) W2X_DPP1("%5", CTRL)
__device__ __forceinline__ void group_sum32_x6(float& a, float& b, float& c, float& d, float& e, float& f) {
    float ta, tb, tc, td, te, tf;
    asm volatile(
        "s_nop 2\n\t" W2X_DPP6("quad_perm:[1,0,3,2]") W2X_DPP6("quad_perm:[2,3,0,1]") W2X_DPP6("row_half_mirror") W2X_DPP6("row_mirror")
        "v_mov_b32 %6, %0\n\tv_mov_b32 %7, %1\n\tv_mov_b32 %8, %2\n\tv_mov_b32 %9, %3\n\tv_mov_b32 %10, %4\n\tv_mov_b32 %11, %5\n\t"
        "v_permlane16_swap_b32 %0, %6\n\tv_permlane16_swap_b32 %1, %7\n\tv_permlane16_swap_b32 %2, %8\n\t"
        "v_permlane16_swap_b32 %3, %9\n\tv_permlane16_swap_b32 %4, %10\n\tv_permlane16_swap_b32 %5, %11\n\t"
        "v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %7\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\tv_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "=&v"(ta), "=&v"(tb), "=&v"(tc), "=&v"(td), "=&v"(te), "=&v"(tf));
}
// Row maximum of three independent values at once (k_swinattn96.hip): the chains fill each other's permlane wait states
// and v_max_f32 is used as is.
__device__ __forceinline__ void rows_max3(float& a0, float& a1, float& a2) {
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
// sum over aligned groups of 32 lanes: DPP inside the 16-lane rows, one row swap across
__device__ __forceinline__ float group_sum32(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    float a = v, b = v; swap16(a, b);
    return a + b;
}

// Build switches (tools/ab/attn192_variants.sh):
//   W2X_A192U_RING   weight-fragment registers of a wave (the ring described in the kernel)
//   W2X_A192U_WPC    workgroups per CU the register budget is set for (3: <= 168 registers)
#ifndef W2X_A192U_RING
#define W2X_A192U_RING 6
#endif
#ifndef W2X_A192U_WPC
#define W2X_A192U_WPC 3
#endif
#ifndef W2X_A192U_EXP
#define W2X_A192U_EXP 0      // timing experiments (wrong results): 1 = leave after the head loop, 2 = skip the head loop, 3 = rows fetched from one cached 64 KB region
#endif
#ifndef W2X_A192U_BIAS_EARLY
#define W2X_A192U_BIAS_EARLY 1   // measured (profiles/r4_kernels/attn192u_*.txt): 0 -> 0.358 ms per 120 x 120 launch, 1 -> 0.308 - 0.322, 2 -> 0.331 (spills)
#endif
constexpr int C = 192, HD = 32, NH = 6, NTOK = 36, G = 2, R = G * NTOK, RT = 5, RP = RT * 16;
constexpr int SLAB = 48, RPX = G * SLAB;       // slab rows per window / in the tile
constexpr int LDX = C + 8;                     // 200 halves: 400-byte rows, 16-byte pieces rotate over the banks
constexpr int XS = RPX * LDX;                  // the slab area; later the head-output tile [RP][LDX], later the output tile [RP][LDX]
constexpr int NPAD = G * 12;                   // slab rows between the left-over tokens (kept at zero)
constexpr int DUMMY = XS * 2;                  // byte offset of one spare row behind the slabs: target of the stores of idle lanes
constexpr int PIX_OFF = DUMMY + LDX * 2;
constexpr int BQ_OFF = PIX_OFF + (R + NPAD) * 8;                // q / k / v bias [3 * C] fp32
constexpr int SMEM192U = BQ_OFF + 3 * C * 4;
constexpr int LPR = 32, PPR = C / 8, RPP = 256 / LPR, NPASS = R / RPP;   // row passes: 32 lanes per row, 8 rows per pass, 9 passes
static_assert(R % RPP == 0, "row passes");
static_assert(RP * LDX <= XS, "the head-output tile and the output tile fit where the slabs were");
static_assert(SMEM192U * W2X_A192U_WPC <= 160 * 1024, "LDS per CU");

__device__ __forceinline__ int slab_row(int t) { return t < 32 ? t : 32 + 4 * (t - 32); }

__global__ __launch_bounds__(256, W2X_A192U_WPC) void swin_attn192u_kernel(const SwinAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;              // [RPX][LDX] normalised x slabs; then the head outputs [RP][LDX] (token order); then the output tile [RP][LDX]
    int2v* Pix = (int2v*)(smem + PIX_OFF);       // [R] {byte offset of the token row's pixel in x / y (kNoRow: none), byte offset of its slab row}, then [NPAD] {-, pad row}

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const int iw0 = blockIdx.x * G, iw1 = iw0 + 1, itotal = p.B * p.nwin;
    const bool wok0 = iw0 < itotal, wok1 = iw1 < itotal;
    const int HW = p.nwin * NTOK;
    const int wb0 = iw0 / p.nwin, wb1 = iw1 / p.nwin;
    const int pixbase0 = wb0 * HW, pixbase1 = wb1 * HW;
    const int wl0 = iw0 - wb0 * p.nwin, wl1 = iw1 - wb1 * p.nwin;
    const int nwx = p.W / 6;
    const int wy0 = wl0 / nwx, wx0 = wl0 - wy0 * nwx, wy1 = wl1 / nwx, wx1 = wl1 - wy1 * nwx;
    const unsigned xbytes = (unsigned)p.B * (unsigned)HW * (C * 2);
    const __amdgpu_buffer_rsrc_t X = make_rsrc(p.x, xbytes), Y = make_rsrc(p.y, xbytes);
    const _Float16* __restrict__ Wqkv = (const _Float16*)p.wqkv_frag;    // [36 row tiles][6 k-steps][64 lanes][8]
    const _Float16* __restrict__ Wproj = (const _Float16*)p.wproj_frag;  // [12 row tiles][6 k-steps][64 lanes][8]
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half8 zero8 = {};

    // this wave's three (window, head) units, in the order they run: head hA on window 0, head hA on window 1, head hC on window wC
    // (git 9576837:tools/ab/k_swinattn192_r3.hip's assignment, so that the two kernels agree bit for bit unit by unit)
    const int hA = wv, hC = 4 + (wv >> 1), wC = wv & 1;
    const int amask0 = wok0 ? p.maskid[wl0] : 0, amask1 = wok1 ? p.maskid[wl1] : 0;

    // Weight fragments reach the MFMAs through a ring of RING registers.  A unit consumes 36 fragments - per k-step: q tile 0, k tile 0, q tile 1,
    // k tile 1 (24), then per k-step v tile 0, v tile 1 (12) - and the wave's three units 108 in a row; fragment i + RING is requested from L2
    // right after the last product that used the register of fragment i, across pass and unit boundaries and under the softmax phases.
    constexpr int RING = W2X_A192U_RING, NFRAG = 108;
    const __amdgpu_buffer_rsrc_t WQ = make_rsrc(Wqkv, 3u * C * C * 2u), WP = make_rsrc(Wproj, (unsigned)C * C * 2u);
    const unsigned wl16 = lane * 16u;
    auto wfrag = [&](int q) {                   // q = unit * 36 + (j < 24: k-step * 4 + feature tile * 2 + (0 q | 1 k);  else 24 + k-step * 2 + feature tile)
        const int u = q / 36, j = q - u * 36;
        const int H = u < 2 ? hA : hC;
        const int M = j < 24 ? (j & 1) : 2, ks = j < 24 ? j >> 2 : (j - 24) >> 1, ft = j < 24 ? (j >> 1) & 1 : (j - 24) & 1;
        return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WQ, wl16, (unsigned)((M * NH + H) * 12 + ft * 6 + ks) * 1024u, 0));
    };
    float4v bq_stage = zero4;                   // requested before the ring's first fragments: the oldest load, nothing waits behind it
    if (tid < 3 * C / 4) bq_stage = *(const float4v*)(p.bqkv + tid * 4);
    const float* Bq = (const float*)(smem + BQ_OFF);
    half8 wr[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) wr[i] = wfrag(i);
#define W2X_RING_NEXT(Q) do { if ((Q) + RING < NFRAG) { wr[(Q) % RING] = wfrag((Q) + RING); asm volatile("" ::: "memory"); } } while (0)

    // ---- source pixel and slab row of every token row (and the pad rows), worked out once per workgroup (one thread per row)
    if (tid < R + NPAD) {
        int pix = -1, srow;
        if (tid >= R) {             // the 12 rows between tokens 32..35 of each slab
            const int k = tid - R, w = k >= 12 ? 1 : 0, kk = k - 12 * w;
            srow = (w * SLAB + 33 + (kk / 3) * 4 + (kk % 3)) * LDX * 2;
        } else {
            const int w = tid >= NTOK ? 1 : 0, t = tid - w * NTOK;
            srow = (w * SLAB + slab_row(t)) * LDX * 2;
            if (w == 0 ? wok0 : wok1) {
                if (p.ry >= 0) {
                    const int ty = t / 6, tx = t - ty * 6;
                    int y = (w == 0 ? wy0 : wy1) * 6 + ty + p.ry, x = (w == 0 ? wx0 : wx1) * 6 + tx + p.rx;
                    y -= y >= p.H ? p.H : 0; x -= x >= p.W ? p.W : 0;
                    pix = (w == 0 ? pixbase0 : pixbase1) + y * p.W + x;
                } else pix = (w == 0 ? pixbase0 : pixbase1) + p.table[(w == 0 ? wl0 : wl1) * NTOK + t];
            }
        }
        Pix[tid] = (int2v){pix < 0 ? (int)kNoRow : (int)((unsigned)pix * (unsigned)(C * 2)), srow};   // offsets are unsigned 32-bit (up to 4 GB per run)
    }
    if (tid < 3 * C / 4) *(float4v*)(smem + BQ_OFF + tid * 16) = bq_stage;
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
            // rows that do not exist and the eight idle lanes of a row read zeros
            xr[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, __builtin_elementwise_add_sat(W2X_A192U_EXP == 3 ? (unsigned)pr[0] & 0xFFFFu : (unsigned)pr[0], lane_off), 0, 0));
        }
        static_assert(NPASS == 9, "the row sums are reduced three passes at a time");
        float sm[NPASS], sq[NPASS];
#pragma unroll
        for (int pg = 0; pg < NPASS; pg += 3) {
#pragma unroll
            for (int ps = pg; ps < pg + 3; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
            group_sum32_x6(sm[pg], sq[pg], sm[pg + 1], sq[pg + 1], sm[pg + 2], sq[pg + 2]);
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const float mean = sm[ps] * (1.f / C);
            const float rstd = __builtin_amdgcn_rsqf(fmaxf(sq[ps] * (1.f / C) - mean * mean, 0.f) + p.eps);   // the argument is >= eps
            // unconditional store: the idle lanes of a row write to the dummy row
            *(half8*)(smem + (li < PPR ? srow[ps] + li * 16 : DUMMY)) = norm8(xr[ps], rstd, -mean * rstd);
        }
        // the 12 rows between tokens 32..35 of each slab are multiplied like the rest (results ignored): keep them finite
#pragma unroll
        for (int k = 0; k < (NPAD + RPP - 1) / RPP; ++k) {
            const int pr = k * RPP + rsub;
            if (pr < NPAD) *(half8*)(smem + (li < PPR ? Pix[R + pr][1] + li * 16 : DUMMY)) = zero8;
        }
    }
    __syncthreads();
#ifdef W2X_A192U_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif

    const float qscale = p.scale * 1.44269504088896341f;   // log2(e) folded into q: softmax uses exp2
    const int lane2 = g * 16 + (fr >> 2);                  // bias-table lane of the query this lane holds in query tile 2

    // head outputs of this wave's three units, as they will go to the head-output tile: [unit][feature tile][query tile] = 4 consecutive features of one query
    half4 oh[3][2][3];

#pragma unroll
    for (int u = 0; u < (W2X_A192U_EXP == 2 ? 0 : 3); ++u) {
        const int w = u == 2 ? wC : u, h = u == 2 ? hC : hA;
        const int amask = w == 0 ? amask0 : amask1;
        const _Float16* xs = Xs + w * SLAB * LDX;
        half8 qf[3], kf[3], vf0[2], vf1[2];
        // the unit's rel-pos bias (+ mask), the initial accumulators of its score products: 27 registers per lane.  W2X_A192U_BIAS_EARLY = 2 requests them
        // here (they travel under the q / k products), 1 behind those products (under the v products), 0 where the score products start
        float4v s[3][3];
        float b2[3];
#define W2X_LOAD_BIAS() {                                                                                   \
            const float* bias = p.bias32 + ((size_t)amask * NH + h) * (3 * 576);                            \
            _Pragma("unroll") for (int qi = 0; qi < 3; ++qi) {                                              \
                const int bl = qi < 2 ? lane : lane2;                                                       \
                s[qi][0] = *(const float4v*)(bias + qi * 576 + bl * 4);                                     \
                s[qi][1] = *(const float4v*)(bias + qi * 576 + 256 + bl * 4);                               \
                s[qi][2] = zero4;                                                                           \
                b2[qi] = bias[qi * 576 + 512 + bl];        /* key tile 2 holds one key per lane: added after the product */ \
            }                                                                                               \
            asm volatile("" ::: "memory");                 /* the requests stay where they are written */  \
        }
#if W2X_A192U_BIAS_EARLY == 2
        W2X_LOAD_BIAS()
#endif
        // ---- q^T (bias = initial accumulator) and k^T (no bias): rows = features (A = weights), columns = slab rows (B = x); tile tt of the result
        //      = [feature tile 0 | feature tile 1] of 16 slab rows
        {
            const float4v bq0 = *(const float4v*)(Bq + h * HD + g * 4), bq1 = *(const float4v*)(Bq + h * HD + 16 + g * 4);
            float4v aq[2][3] = {{bq0, bq0, bq0}, {bq1, bq1, bq1}}, ak[2][3] = {{zero4, zero4, zero4}, {zero4, zero4, zero4}};
            half8 xf[2][3];                // x fragments of k-step ks in xf[ks & 1], requested one k-step ahead
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) xf[0][tt] = *(const half8*)(xs + (tt * 16 + fr) * LDX + g * 8);
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                const int ks = j >> 2, ft = (j >> 1) & 1, isk = j & 1, q = u * 36 + j;
                if ((j & 3) == 0 && ks + 1 < 6) {
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) xf[(ks + 1) & 1][tt] = *(const half8*)(xs + (tt * 16 + fr) * LDX + (ks + 1) * 32 + g * 8);
                }
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) {
                    if (isk) ak[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[q % RING], xf[ks & 1][tt], ak[ft][tt], 0, 0, 0);
                    else aq[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[q % RING], xf[ks & 1][tt], aq[ft][tt], 0, 0, 0);
                }
                W2X_RING_NEXT(q);
            }
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const float4v a0 = aq[0][tt] * qscale, a1 = aq[1][tt] * qscale;
#pragma unroll
                for (int j = 0; j < 4; ++j) { qf[tt][j] = (_Float16)a0[j]; qf[tt][4 + j] = (_Float16)a1[j]; kf[tt][j] = (_Float16)ak[0][tt][j]; kf[tt][4 + j] = (_Float16)ak[1][tt][j]; }
            }
        }
#if W2X_A192U_BIAS_EARLY == 1
        W2X_LOAD_BIAS()
#endif
        // ---- v: rows = slab rows (A = x), columns = features (B = weights); vf0 = tokens 0..31, vf1 = slab row 32 + 4g = token 32 + g
        {
            float4v av[3][2] = {{zero4, zero4}, {zero4, zero4}, {zero4, zero4}};
            half8 xf[2][3];
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) xf[0][tt] = *(const half8*)(xs + (tt * 16 + fr) * LDX + g * 8);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ks = j >> 1, ft = j & 1, q = u * 36 + 24 + j;
                if (ft == 0 && ks + 1 < 6) {
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) xf[(ks + 1) & 1][tt] = *(const half8*)(xs + (tt * 16 + fr) * LDX + (ks + 1) * 32 + g * 8);
                }
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) av[tt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ks & 1][tt], wr[q % RING], av[tt][ft], 0, 0, 0);
                W2X_RING_NEXT(q);
            }
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                half8 f0, f1 = zero8;
#pragma unroll
                for (int j = 0; j < 4; ++j) { f0[j] = (_Float16)av[0][ft][j]; f0[4 + j] = (_Float16)av[1][ft][j]; }
                f1[0] = (_Float16)av[2][ft][0];
                vf0[ft] = f0; vf1[ft] = f1;
            }
        }
#if !W2X_A192U_BIAS_EARLY
        W2X_LOAD_BIAS()
#endif
#undef W2X_LOAD_BIAS
        // ---- S^T = K Q^T on top of the bias, softmax over the keys (lane-local + two row swaps), O^T = V^T P^T scaled by 1/l (git 9576837:tools/ab/k_swinattn192_r3.hip's
        //      attend(), instruction for instruction); the result stays in registers
#pragma unroll
        for (int qi = 0; qi < 3; ++qi)
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) s[qi][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf[qi], s[qi][kt], 0, 0, 0);
        float mx[3];
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            s[qi][2][0] += b2[qi];
            // plain fmaxf on purpose (the inputs are MFMA results: wait states are only inserted for instructions the compiler sees)
            mx[qi] = fmaxf(fmaxf(fmaxf(fmaxf(s[qi][0][0], s[qi][0][1]), s[qi][0][2]), fmaxf(fmaxf(s[qi][0][3], s[qi][1][0]), s[qi][1][1])), fmaxf(fmaxf(s[qi][1][2], s[qi][1][3]), s[qi][2][0]));
        }
        rows_max3(mx[0], mx[1], mx[2]);
        half8 pf0[3], pf1[3];
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            half8 f;
            const float2v m2 = {mx[qi], mx[qi]};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {   // the subtractions as v_pk_add_f32
                const float2v d0 = (float2v){s[qi][kt][0], s[qi][kt][1]} - m2, d1 = (float2v){s[qi][kt][2], s[qi][kt][3]} - m2;
                f[4 * kt + 0] = (_Float16)__builtin_amdgcn_exp2f(d0[0]); f[4 * kt + 1] = (_Float16)__builtin_amdgcn_exp2f(d0[1]);
                f[4 * kt + 2] = (_Float16)__builtin_amdgcn_exp2f(d1[0]); f[4 * kt + 3] = (_Float16)__builtin_amdgcn_exp2f(d1[1]);
            }
            pf0[qi] = f;
            half8 t = zero8; t[0] = (_Float16)__builtin_amdgcn_exp2f(s[qi][2][0] - mx[qi]);
            pf1[qi] = t;
        }
        float inv[3];
        const float4v bv[2] = {*(const float4v*)(Bq + 2 * C + h * HD + g * 4), *(const float4v*)(Bq + 2 * C + h * HD + 16 + g * 4)};
        {
            const _Float16 one = (_Float16)1.f;
            const half8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                float4v l = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf0[qi], zero4, 0, 0, 0);
                l = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf1[qi], l, 0, 0, 0);
                inv[qi] = __builtin_amdgcn_rcpf(l[0]);
            }
        }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                float4v o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf0[ft], pf0[qi], zero4, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf1[ft], pf1[qi], o, 0, 0, 0);
                const float2v i2 = {inv[qi], inv[qi]};
                const float2v o0 = __builtin_elementwise_fma((float2v){o[0], o[1]}, i2, (float2v){bv[ft][0], bv[ft][1]});
                const float2v o1 = __builtin_elementwise_fma((float2v){o[2], o[3]}, i2, (float2v){bv[ft][2], bv[ft][3]});
                oh[u][ft][qi] = (half4){(_Float16)o0[0], (_Float16)o0[1], (_Float16)o1[0], (_Float16)o1[1]};
                // pinned: left to itself the compiler keeps the four fp32 values and converts at the store after the barrier (24 more registers per
                // unit, spilled at three workgroups per CU)
                asm volatile("" : "+v"(oh[u][ft][qi]));
            }
    }
#undef W2X_RING_NEXT
#if W2X_A192U_EXP == 1
    if (p.B >= 0) { if (tid == 0) ((float*)p.y)[blockIdx.x] = (float)oh[0][0][0][0] + (float)oh[1][1][1][1] + (float)oh[2][0][2][2]; return; }
#endif
#if W2X_A192U_EXP == 2
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) oh[u][ft][qi] = (half4){};
#endif
#ifdef W2X_A192U_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // the projection's weight fragments of the first k-step and its bias vectors: requested before the barriers, they are here when the products start
    half8 wp[2][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) wp[0][t] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WP, wl16, (unsigned)((wv * 3 + t) * 6 + 0) * 1024u, 0));
    float4v bp[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bp[t] = *(const float4v*)(p.bproj + (wv * 3 + t) * 16 + g * 4);
    __syncthreads();      // nobody reads the slabs any more: the head outputs go where they were

    _Float16* Os = Xs;    // [RP][LDX], token order
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int w = u == 2 ? wC : u, h = u == 2 ? hC : hA;
        const bool ok = w == 0 ? wok0 : wok1;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                const int query = qi < 2 ? qi * 16 + fr : 32 + (fr >> 2);
                if (ok && (qi < 2 || (fr & 3) == 0)) *(half4*)(Os + (w * NTOK + query) * LDX + h * HD + ft * 16 + g * 4) = oh[u][ft][qi];
            }
    }
    // rows 72..79 of the head-output tile belong to no token: the projection multiplies them like the rest (results never stored): keep them finite
    if (tid < (RP - R) * (C / 8)) *(half8*)(Os + (R + tid / (C / 8)) * LDX + (tid % (C / 8)) * 8) = zero8;
    // windows that do not exist (the last workgroup of a run) leave their rows of the tile as the slabs were: finite values, never stored
    __syncthreads();

    // The residual rows (the same pixels again, for y = x + ...) are requested here, in front of the projection (git 9576837:tools/ab/k_swinattn192_r3.hip): the
    // fetch travels under its 90 products.
    const int li_r = tid & (LPR - 1), rsub_r = tid / LPR;
    half8 xres[NPASS];
    unsigned my_off[NPASS];
    {
        const unsigned lane_off = li_r < PPR ? li_r * 16u : kNoRow;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            my_off[ps] = __builtin_elementwise_add_sat((unsigned)Pix[ps * RPP + rsub_r][0], lane_off);
            xres[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, my_off[ps], 0, W2X_LD_LAST_AUX));
        }
        asm volatile("" ::: "memory");      // keeps the requests here (the scheduler would sink them to their use behind the projection)
    }

    // ---- proj, transposed: out^T = Wproj Os^T + b (rows = output channels, columns = tokens), so a lane ends with 4 consecutive
    // channels of one token (bias as the initial accumulator, 8-byte LDS stores).  Wave w owns output channels 48w .. 48w+47 for all
    // five row tiles; k-step outermost: three weight fragments per k-step (requested a k-step ahead), the 15 accumulators stay in registers.
    float4v acc[RT][3];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[mt][t] = bp[t];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        if (ks + 1 < 6) {
#pragma unroll
            for (int t = 0; t < 3; ++t) wp[(ks + 1) & 1][t] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WP, wl16, (unsigned)((wv * 3 + t) * 6 + ks + 1) * 1024u, 0));
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
            const half8 of = *(const half8*)(Os + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[ks & 1][t], of, acc[mt][t], 0, 0, 0);
        }
    }
    __syncthreads();      // every wave has read the head outputs: the output tile goes over them
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int t = 0; t < 3; ++t)
            *(half4*)(Xs + (mt * 16 + fr) * LDX + (wv * 3 + t) * 16 + g * 4) = (half4){(_Float16)acc[mt][t][0], (_Float16)acc[mt][t][1], (_Float16)acc[mt][t][2], (_Float16)acc[mt][t][3]};
    __syncthreads();

    // ---- row pieces: + residual x, scatter store, LayerNorm statistics for the next op
    {
        const int li = li_r, rsub = rsub_r;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + rsub;
            const bool ok = my_off[ps] != kNoRow;
            // (idle lanes read the next row's first pieces: their store is dropped)
            half8 o = *(const half8*)(Xs + r * LDX + li * 8) + xres[ps];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), Y, my_off[ps], 0, W2X_ST_AUX);
            if (p.stats_out) {
                if (!ok) o = half8{};
                const size_t pix = my_off[ps] / (C * 2);
                float s, q;
                sum_sq8(o, s, q);
                s = group_sum32(s);
                q = group_sum32(q);
                const float mean = s * (1.f / C);
                if (ok && li == 0) { p.stats_out[2 * pix] = mean; p.stats_out[2 * pix + 1] = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out); }
            }
        }
    }
}

}  // namespace

// same contract as launch_swin_attn192 (git 9576837:tools/ab/k_swinattn192_r3.hip)
hipError_t launch_swin_attn192u(const SwinAttnParams& p, hipStream_t s) {
    auto kern = swin_attn192u_kernel;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, SMEM192U, lds_ok); e != hipSuccess) return e;
    // the kernel addresses x / y with 32-bit byte offsets: passes beyond that are cut into runs of whole images (k_swinattn96.hip)
    const size_t img_bytes = (size_t)p.nwin * NTOK * C * 2;
    if (img_bytes == 0 || img_bytes > kMaxBufBytes) return hipErrorInvalidValue;
    const int per_run = (int)std::min<size_t>((size_t)p.B, kMaxBufBytes / img_bytes);
    for (int b0 = 0; b0 < p.B; b0 += per_run) {
        SwinAttnParams q = p;
        q.B = std::min(per_run, p.B - b0);
        q.x = (const char*)p.x + (size_t)b0 * img_bytes;
        q.y = (char*)p.y + (size_t)b0 * img_bytes;
        if (p.stats_out) q.stats_out = p.stats_out + (size_t)b0 * p.nwin * NTOK * 2;
        const long total_win = (long)q.B * q.nwin;
        hipLaunchKernelGGL(kern, dim3((unsigned)((total_win + G - 1) / G)), dim3(256), SMEM192U, s, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace w2x
