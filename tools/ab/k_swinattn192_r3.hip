// Fused Swin attention branch, C = 192 / 6 heads of 32 / window 6x6, register-resident variant for gfx950.
//     y = x + proj( W-MSA( LayerNorm(x) ) )
// Same math, parameters and bias-table layout as k_swinattn96.hip, and the same unit mapping: a workgroup = 4 waves = 2 windows,
// wave v takes head v on BOTH windows (a head's weights are read once for two windows, every fragment feeds six MFMAs) and then head
// 4 + (v >> 1) on window v & 1.  With a head dimension of 32 the accumulator layout of v_mfma_f32_16x16x32_f16 (lane (col, g)
// holds rows 4g..4g+3) IS an operand layout (lane (row|col, g) holds 8 k-values) once two 16-feature tiles are paired, with the
// k order permuted the same way on both sides.  So
//     q^T, k^T (rows = features) -> B / A operands of S^T = K Q^T      straight from the accumulators,
//     v (rows = tokens)          -> A operand of O^T = V^T P^T         straight from the accumulators,
//     S^T after the softmax      -> B operand of O^T                   (as in the other variants),
// and q, k, v, S, P never touch LDS.  LDS only holds the normalised x slabs (MFMA operand of the q/k/v products, read by
// all waves) and the head outputs (operand of proj).  Weights come from L2 as fragments through a ring of eight registers, each
// fragment requested eight fragments ahead of its use across matrix and unit boundaries; x fragments are requested one k-step
// ahead.  Four workgroup barriers in the whole kernel; the vector work around the products is trimmed as in k_swinattn96.hip
// (ones-operand MFMA for the softmax denominators, folded k / v biases, q bias as initial accumulator, packed fp32, buffer loads /
// stores and v_fma_mix in the row phases, transposed proj).  At 234 VGPRs = two workgroups per CU the kernel is latency-bound:
// ring depths of 6 / 8 / 12 measure the same, a third workgroup does not fit LDS (70 KB each).
//
// Window slab: 48 rows; tokens 0..31 on rows 0..31, tokens 32..35 on rows 32, 36, 40, 44, zero rows between.  That puts
// key 32+g on row 4g of the third key tile (one per lane group, register j = 0: the softmax touches 9 instead of 12
// values per lane, the layout the bias table is stored for) and - because q, k and v are all computed from the same slab -
// needs no data movement for it: the third tiles of K, V and Q come out of the MFMAs already arranged that way.
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
//   W2X_A192_BQ_LDS      the q / k / v bias vectors (576 floats) are copied to LDS when the workgroup starts and read from there.  As global loads
//                        they were requested where they are used - the q bias as the initial accumulator of the first q product of a head, the v
//                        bias inside every unit - and each was a round trip to L2 in front of a waiting wave (two waves per SIMD here).
//   W2X_A192_BIAS_AHEAD  the rel-pos bias (+ mask) values of a unit (27 registers per lane, the initial accumulators of its score products) are
//                        requested a phase ahead - before the v products for a head's first unit, under the first unit's softmax for the second -
//                        instead of at the top of the unit.
//   W2X_A192_STAGGER=n   experiment: the workgroups of the launch's first generation that land on the second wave slot of their SIMD sleep n x 8128
//                        cycles first, so that the two workgroups of a CU run half a period apart (one in its row phases while the other multiplies)
//                        instead of in lockstep; later generations inherit the offset.
//   W2X_A192_PRIO=m      s_setprio by phase.  1: the head loop at priority 1, row phases at 0;  2: priority rises with progress (0 gather,
//                        1 head loop, 2 projection, 3 final rows) - the older workgroup of a CU wins every arbitration and leaves sooner;
//                        3: head loop 2, projection 1, row phases 0;  4: only the softmax / O sections of the head loop at 1;  5: only the q / k / v
//                        products of the head loop at 1 (the softmax / O sections at 0).
#ifndef W2X_A192_RING
#define W2X_A192_RING 8        // weight-fragment registers of a wave (the ring described in the kernel)
#endif
#ifndef W2X_A192_BQ_LDS
#define W2X_A192_BQ_LDS 1
#endif
#ifndef W2X_A192_BIAS_AHEAD
#define W2X_A192_BIAS_AHEAD 1
#endif
constexpr int C = 192, HD = 32, NH = 6, NTOK = 36, G = 2, R = G * NTOK, RT = 5, RP = RT * 16;
constexpr int SLAB = 48, RPX = G * SLAB;       // slab rows per window / in the tile
constexpr int LDX = C + 8;                     // 200 halves: 400-byte rows, 16-byte pieces rotate over the banks
constexpr int XS = RPX * LDX, OS = RP * LDX;
constexpr int NPAD = G * 12;                   // slab rows between the left-over tokens (kept at zero)
constexpr int BQ_OFF = (XS + OS) * 2 + (R + NPAD) * 8;          // q / k / v bias [3 * C] fp32 (W2X_A192_BQ_LDS)
constexpr int SMEM192 = BQ_OFF + 3 * C * 4;
constexpr int DUMMY = XS * 2;                  // byte offset of a row nobody reads at that point (first row of Os): target of the stores of idle lanes
constexpr int LPR = 32, PPR = C / 8, RPP = 256 / LPR, NPASS = R / RPP;   // row passes: 32 lanes per row, 8 rows per pass, 9 passes
static_assert(R % RPP == 0, "row passes");

__device__ __forceinline__ int slab_row(int t) { return t < 32 ? t : 32 + 4 * (t - 32); }

// Diagnostic build only (instantiate the kernel with STAMPS = true): per-phase s_memtime deltas summed over all waves.
// g_sa192_stamps[k]: 0 gather+LN+barrier, 1 q and k products, 2 v products, 3 S + softmax, 4 O + store, 5 barrier + proj,
// 6 barrier + final rows, 7 waves counted.
__device__ unsigned long long g_sa192_stamps[8];
#define W2X_STAMP(K)                                                                                        \
    if (STAMPS) {                                                                                           \
        unsigned long long t_;                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (lane == 0) atomicAdd(&g_sa192_stamps[K], t_ - tprev);                                           \
        tprev = t_;                                                                                         \
    }

template <bool STAMPS>
__global__ __launch_bounds__(256, 2) void swin_attn192_kernel(const SwinAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;              // [RPX][LDX] normalised x slabs; later the output tile [RP][LDX]
    _Float16* Os = Xs + XS;                      // [RP][LDX]  attention output, all heads, token order
    int2v* Pix = (int2v*)(Os + OS);              // [R] {byte offset of the token row's pixel in x / y (kNoRow: none), byte offset of its slab row}, then [NPAD] {-, pad row}

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
#ifdef W2X_A192_STAGGER
    if (blockIdx.x < 2u * 256u) {      // HW_REG_HW_ID bits 3:0 = wave slot on the SIMD
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (slot & 1) for (int i = 0; i < W2X_A192_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    unsigned long long tprev = 0;
    if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); if (lane == 0) atomicAdd(&g_sa192_stamps[7], 1ull); }

    // this wave's three (window, head) units: head hA on both windows, head hC on window wC (k_swinattn96.hip's mapping): a head's
    // weights are read once for two windows, every fragment feeds 6 MFMAs there instead of 3
    const int hA = wv, hC = 4 + (wv >> 1), wC = wv & 1;
    const int amask0 = wok0 ? p.maskid[wl0] : 0, amask1 = wok1 ? p.maskid[wl1] : 0;

    // Weight fragments reach the MFMAs through a ring of RING registers.  The six 32 x 192 matrices of a wave (q, k, v of head hA,
    // then of head hC) are 72 fragments in consumption order (matrix, k-step, feature tile); fragment i+RING is requested from L2
    // right after the last MFMA that used the register of fragment i - across matrix boundaries and across the softmax phases -
    // so a load has RING x 6 (or 3) MFMAs and everything between the products to land.  (Before: two alternating sets of 12
    // registers, 96 VGPRs, which the scheduler partly sank to the consumers anyway.)
    constexpr int RING = W2X_A192_RING, NFRAG = 72;
    // (through buffer resources: the lane part of a fragment address is lane * 16 for every fragment, the rest is scalar arithmetic -
    // as global loads each fragment cost a 64-bit vector add)
    const __amdgpu_buffer_rsrc_t WQ = make_rsrc(Wqkv, 3u * C * C * 2u), WP = make_rsrc(Wproj, (unsigned)C * C * 2u);
    const unsigned wl16 = lane * 16u;
    auto wfrag = [&](int q) {                   // q = matrix * 12 + k-step * 2 + feature tile
        const int mat = q / 12, j = q - mat * 12, M = mat % 3, ks = j >> 1, ft = j & 1;
        const int H = mat < 3 ? hA : hC;
        return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WQ, wl16, (unsigned)((M * NH + H) * 12 + ft * 6 + ks) * 1024u, 0));
    };
#if W2X_A192_BQ_LDS
    float4v bq_stage = zero4;                   // requested before the ring's first fragments: the oldest load, nothing waits behind it
    if (tid < 3 * C / 4) bq_stage = *(const float4v*)(p.bqkv + tid * 4);
    const float* Bq = (const float*)(smem + BQ_OFF);
#else
    const float* Bq = p.bqkv;
#endif
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
#if W2X_A192_BQ_LDS
    if (tid < 3 * C / 4) *(float4v*)(smem + BQ_OFF + tid * 16) = bq_stage;
#endif
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
            xr[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, __builtin_elementwise_add_sat((unsigned)pr[0], lane_off), 0, 0));
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
    W2X_STAMP(0)
#if defined(W2X_A192_PRIO) && W2X_A192_PRIO == 3
    __builtin_amdgcn_s_setprio(2);
#elif defined(W2X_A192_PRIO) && W2X_A192_PRIO != 4
    __builtin_amdgcn_s_setprio(1);
#endif

    const float qscale = p.scale * 1.44269504088896341f;   // log2(e) folded into q: softmax uses exp2
    const int lane2 = g * 16 + (fr >> 2);                  // bias-table lane of the query this lane holds in query tile 2
    half8 wp[3][6];                                        // proj fragments of this wave's three n-tiles, fetched under the last unit

    // ---- one (window, head) unit from its q / k / v fragments: S^T = K Q^T on top of the bias, softmax over the keys (lane-local +
    // two row swaps), O^T = V^T P^T scaled by 1/l and parked in Os (token order).  The denominators come off the matrix pipe (a
    // ones matrix in place of V^T, see k_swinattn96.hip); the k bias drops out of the softmax, the v bias is added to the
    // normalised output.
    struct UnitBias { float4v s[3][2]; float b2[3]; };        // rel-pos bias (+ mask) of one unit in the score accumulators' layout
    auto load_bias = [&](const int amask, const int h, UnitBias& ub) {
        const float* bias = p.bias32 + ((size_t)amask * NH + h) * (3 * 576);
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            const int bl = qi < 2 ? lane : lane2;
            ub.s[qi][0] = *(const float4v*)(bias + qi * 576 + bl * 4);
            ub.s[qi][1] = *(const float4v*)(bias + qi * 576 + 256 + bl * 4);
            ub.b2[qi] = bias[qi * 576 + 512 + bl];        // key tile 2 holds one key per lane: added after the product
        }
        asm volatile("" ::: "memory");                     // the requests stay where they are written
    };
    // `ub` holds this unit's bias (requested earlier with W2X_A192_BIAS_AHEAD, else here); next_amask >= 0: the next unit of the same head is
    // requested into `ub` again once the score products have taken this one's values
    auto attend = [&](const int w, const int h, const bool ok, const int amask, const half8 (&qf)[3], const half8 (&kf)[3], const half8 (&vf0)[2], const half8 (&vf1)[2], const bool last,
                      UnitBias& ub, const int next_amask) {
#if defined(W2X_A192_PRIO) && W2X_A192_PRIO == 4
        __builtin_amdgcn_s_setprio(1);
#elif defined(W2X_A192_PRIO) && W2X_A192_PRIO == 5
        __builtin_amdgcn_s_setprio(0);
#endif
        float4v s[3][3];
        float b2[3];
#if !W2X_A192_BIAS_AHEAD
        load_bias(amask, h, ub);
#endif
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) { s[qi][0] = ub.s[qi][0]; s[qi][1] = ub.s[qi][1]; s[qi][2] = zero4; b2[qi] = ub.b2[qi]; }
#pragma unroll
        for (int qi = 0; qi < 3; ++qi)
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) s[qi][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf[qi], s[qi][kt], 0, 0, 0);
#if W2X_A192_BIAS_AHEAD
        if (next_amask >= 0) load_bias(next_amask, h, ub);
#endif
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
        if (last) {   // fetch this wave's proj fragments under the last products
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) wp[t][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WP, wl16, (unsigned)((wv * 3 + t) * 6 + ks) * 1024u, 0));
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
                const int query = qi < 2 ? qi * 16 + fr : 32 + (fr >> 2);
                if (ok && (qi < 2 || (fr & 3) == 0)) {
                    const float2v i2 = {inv[qi], inv[qi]};
                    const float2v o0 = __builtin_elementwise_fma((float2v){o[0], o[1]}, i2, (float2v){bv[ft][0], bv[ft][1]});
                    const float2v o1 = __builtin_elementwise_fma((float2v){o[2], o[3]}, i2, (float2v){bv[ft][2], bv[ft][3]});
                    const half4 oh = {(_Float16)o0[0], (_Float16)o0[1], (_Float16)o1[0], (_Float16)o1[1]};
                    *(half4*)(Os + (w * NTOK + query) * LDX + h * HD + ft * 16 + g * 4) = oh;
                }
            }
#if defined(W2X_A192_PRIO) && W2X_A192_PRIO == 4
        __builtin_amdgcn_s_setprio(0);
#elif defined(W2X_A192_PRIO) && W2X_A192_PRIO == 5
        __builtin_amdgcn_s_setprio(1);
#endif
    };
    // accumulators of one window -> operand fragments: q^T / k^T (rows = features): tile tt = [feature tile 0 | feature tile 1];
    // v (rows = slab rows): vf0 = tokens 0..31, vf1 = slab row 32 + 4g = token 32 + g
    auto pack_qk = [&](const float4v (&a)[2][3], const bool is_q, half8 (&f)[3]) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const float4v a0 = is_q ? a[0][tt] * qscale : a[0][tt], a1 = is_q ? a[1][tt] * qscale : a[1][tt];
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[tt][j] = (_Float16)a0[j]; f[tt][4 + j] = (_Float16)a1[j]; }
        }
    };
    auto pack_v = [&](const float4v (&a)[3][2], half8 (&vf0)[2], half8 (&vf1)[2]) {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            half8 f0, f1 = zero8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { f0[j] = (_Float16)a[0][ft][j]; f0[4 + j] = (_Float16)a[1][ft][j]; }
            f1[0] = (_Float16)a[2][ft][0];
            vf0[ft] = f0; vf1[ft] = f1;
        }
    };

    UnitBias ubias;
    // ---- head hA on both windows: every weight fragment multiplies six token tiles
#define W2X_LOAD_X2(DST, KS_)                                                                                                   \
    _Pragma("unroll") for (int w_ = 0; w_ < 2; ++w_) _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_)                           \
        DST[w_][t_] = *(const half8*)(Xs + (w_ * SLAB + t_ * 16 + fr) * LDX + (KS_) * 32 + g * 8);
    {
        half8 qf[2][3], kf[2][3], vf0[2][2], vf1[2][2];
        const float4v bq0 = *(const float4v*)(Bq + hA * HD + g * 4), bq1 = *(const float4v*)(Bq + hA * HD + 16 + g * 4);
#pragma unroll
        for (int m = 0; m < 2; ++m) {      // q^T (bias = initial accumulator) and k^T (no bias): rows = features (A = weights), columns = slab rows (B = x)
            const float4v b0 = m == 0 ? bq0 : zero4, b1 = m == 0 ? bq1 : zero4;
            float4v a[2][2][3] = {{{b0, b0, b0}, {b1, b1, b1}}, {{b0, b0, b0}, {b1, b1, b1}}};
            half8 xf[2][2][3];             // x fragments of k-step ks in xf[ks & 1], requested one k-step ahead
            W2X_LOAD_X2(xf[0], 0);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ks = j >> 1, ft = j & 1, q = m * 12 + j;
                if (ft == 0 && ks + 1 < 6) W2X_LOAD_X2(xf[(ks + 1) & 1], ks + 1);
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) a[w][ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[q % RING], xf[ks & 1][w][tt], a[w][ft][tt], 0, 0, 0);
                W2X_RING_NEXT(q);
            }
            pack_qk(a[0], m == 0, m == 0 ? qf[0] : kf[0]);
            pack_qk(a[1], m == 0, m == 0 ? qf[1] : kf[1]);
        }
        W2X_STAMP(1)
#if W2X_A192_BIAS_AHEAD
        load_bias(amask0, hA, ubias);
#endif
        {                                  // v: rows = slab rows (A = x), columns = features (B = weights)
            float4v a[2][3][2] = {{{zero4, zero4}, {zero4, zero4}, {zero4, zero4}}, {{zero4, zero4}, {zero4, zero4}, {zero4, zero4}}};
            half8 xf[2][2][3];
            W2X_LOAD_X2(xf[0], 0);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ks = j >> 1, ft = j & 1, q = 24 + j;
                if (ft == 0 && ks + 1 < 6) W2X_LOAD_X2(xf[(ks + 1) & 1], ks + 1);
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) a[w][tt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ks & 1][w][tt], wr[q % RING], a[w][tt][ft], 0, 0, 0);
                W2X_RING_NEXT(q);
            }
            pack_v(a[0], vf0[0], vf1[0]);
            pack_v(a[1], vf0[1], vf1[1]);
        }
        W2X_STAMP(2)
        attend(0, hA, wok0, amask0, qf[0], kf[0], vf0[0], vf1[0], false, ubias, amask1);
        attend(1, hA, wok1, amask1, qf[1], kf[1], vf0[1], vf1[1], false, ubias, -1);
        W2X_STAMP(3)
    }
    // ---- head hC on window wC
    {
        half8 qf[3], kf[3], vf0[2], vf1[2];
        const _Float16* xs = Xs + wC * SLAB * LDX;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float4v b0 = m == 0 ? *(const float4v*)(Bq + hC * HD + g * 4) : zero4;
            const float4v b1 = m == 0 ? *(const float4v*)(Bq + hC * HD + 16 + g * 4) : zero4;
            float4v a[2][3] = {{b0, b0, b0}, {b1, b1, b1}};
            half8 xf[3];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ks = j >> 1, ft = j & 1, q = 36 + m * 12 + j;
                if (ft == 0) {
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) xf[tt] = *(const half8*)(xs + (tt * 16 + fr) * LDX + ks * 32 + g * 8);
                }
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) a[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[q % RING], xf[tt], a[ft][tt], 0, 0, 0);
                W2X_RING_NEXT(q);
            }
            pack_qk(a, m == 0, m == 0 ? qf : kf);
        }
#if W2X_A192_BIAS_AHEAD
        load_bias(wC == 0 ? amask0 : amask1, hC, ubias);
#endif
        {
            float4v a[3][2] = {{zero4, zero4}, {zero4, zero4}, {zero4, zero4}};
            half8 xf[3];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int ks = j >> 1, ft = j & 1, q = 60 + j;
                if (ft == 0) {
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) xf[tt] = *(const half8*)(xs + (tt * 16 + fr) * LDX + ks * 32 + g * 8);
                }
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) a[tt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[tt], wr[q % RING], a[tt][ft], 0, 0, 0);
                W2X_RING_NEXT(q);
            }
            pack_v(a, vf0, vf1);
        }
        attend(wC, hC, wC == 0 ? wok0 : wok1, wC == 0 ? amask0 : amask1, qf, kf, vf0, vf1, true, ubias, -1);
        W2X_STAMP(4)
    }
#undef W2X_RING_NEXT
#undef W2X_LOAD_X2
#if defined(W2X_A192_PRIO) && W2X_A192_PRIO == 2
    __builtin_amdgcn_s_setprio(2);
#elif defined(W2X_A192_PRIO) && W2X_A192_PRIO == 3
    __builtin_amdgcn_s_setprio(1);
#elif defined(W2X_A192_PRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();      // every wave's head outputs are in Os; nobody reads the slabs any more

    // The residual rows (the same pixels again, for y = x + ...) are requested here, in front of the projection: its weight fragments
    // were requested long ago (loads return in order, so the products below wait for nothing new), the head loop's registers are
    // free, and the fetch travels under the 90 products of the projection.  (Round 2 requested them after the projection and waited.)
    // (the projection's bias vectors first: anything requested AFTER the rows would wait for them - loads return in order)
    float4v bp[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bp[t] = *(const float4v*)(p.bproj + (wv * 3 + t) * 16 + g * 4);
    const int li_r = tid & (LPR - 1), rsub_r = tid / LPR;
    half8 xres[NPASS];
    unsigned my_off[NPASS];
#ifndef W2X_A192_XRES_LATE
    {
        const unsigned lane_off = li_r < PPR ? li_r * 16u : kNoRow;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            my_off[ps] = __builtin_elementwise_add_sat((unsigned)Pix[ps * RPP + rsub_r][0], lane_off);
            xres[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, my_off[ps], 0, 0));
        }
        asm volatile("" ::: "memory");      // keeps the requests here (the scheduler would sink them to their use behind the projection)
    }
#endif

    // ---- proj, transposed: out^T = Wproj Os^T + b (rows = output channels, columns = tokens), so a lane ends with 4 consecutive
    // channels of one token (bias as the initial accumulator, 8-byte LDS stores).  Wave w owns output channels 48w .. 48w+47 for all
    // five row tiles; the tile goes over Xs in token order.
    {
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
            float4v acc[3] = {bp[0], bp[1], bp[2]};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const half8 of = *(const half8*)(Os + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[t][ks], of, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
                *(half4*)(Xs + (mt * 16 + fr) * LDX + (wv * 3 + t) * 16 + g * 4) = (half4){(_Float16)acc[t][0], (_Float16)acc[t][1], (_Float16)acc[t][2], (_Float16)acc[t][3]};
        }
    }
#if defined(W2X_A192_PRIO) && W2X_A192_PRIO == 2
    __builtin_amdgcn_s_setprio(3);
#elif defined(W2X_A192_PRIO) && W2X_A192_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
    W2X_STAMP(5)

    // ---- row pieces: + residual x, scatter store, LayerNorm statistics for the next op
    {
        const int li = li_r, rsub = rsub_r;
#ifdef W2X_A192_XRES_LATE
        const unsigned lane_off = li < PPR ? li * 16u : kNoRow;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            my_off[ps] = __builtin_elementwise_add_sat((unsigned)Pix[ps * RPP + rsub][0], lane_off);
            xres[ps] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(X, my_off[ps], 0, 0));
        }
#endif
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + rsub;
            const bool ok = my_off[ps] != kNoRow;
            // (idle lanes read the next row's first pieces: their store is dropped)
            half8 o = *(const half8*)(Xs + r * LDX + li * 8) + xres[ps];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), Y, my_off[ps], 0, 0);
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
    W2X_STAMP(6)
}

}  // namespace

hipError_t launch_swin_attn192(const SwinAttnParams& p, hipStream_t s) {
#ifdef W2X_A192_STAMPS
    auto kern = swin_attn192_kernel<true>;    // diagnostic build (tools/ab/attn192_variants.sh): per-phase s_memtime stamps into g_sa192_stamps
#else
    auto kern = swin_attn192_kernel<false>;
#endif
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, SMEM192, lds_ok); e != hipSuccess) return e;
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
        hipLaunchKernelGGL(kern, dim3((unsigned)((total_win + G - 1) / G)), dim3(256), SMEM192, s, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

#ifdef W2X_A192_STAMPS
// reads and clears the phase sums of the stamped build: out[0..6] wave cycles per phase summed over all waves, out[7] waves counted
extern "C" void w2x_sa192_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sa192_stamps), 8 * sizeof(unsigned long long));
    const unsigned long long zero[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sa192_stamps), zero, sizeof(zero));
}
#endif

}  // namespace w2x
