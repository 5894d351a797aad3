// Fused transformer MLP branch, C = 96, resident-weight variant for gfx950 on 32x32 matrix tiles:
//     y = x + W2 * gelu(W1 * LayerNorm(x) + b1) + b2
// The schedule of k_mlp96p.hip (round 2; tools/ab/k_mlp96p.hip keeps it for A/B runs) with v_mfma_f32_32x32x16_f16 in place of
// v_mfma_f32_16x16x32_f16.  Why: the kernel is bound by vector-instruction issue (7 VALU per GELU value), and on one SIMD matrix
// and vector instructions of different waves do NOT overlap for 16x16 tiles - tools/issue_model.hip, three waves per SIMD: a group of
// one 16x16x32 product + 8 VALU costs 27.7 ticks = 12.0 (the product alone) + 15.6 (the VALU alone), while one 32x32x16 product
// (twice the FLOP) + 16 VALU costs 46.7 against 23.4 + 31.2: a third of the matrix time hides behind the vector work.  Same FLOP,
// same LDS traffic (a 1 KiB weight fragment per 32 K FLOP), same registers.
// Per-wave dataflow as in k_mlp2.hip (a wave owns 32 token rows from the first load to the last store,
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

// Timing experiments (wrong results; tools/ab/mlp96_variants.sh): W2X_MLP_EXP bit 0 rows fetched from / stored to one small cached
// region, bit 1 GELU replaced by the bare conversion, bit 2 no matrix products, bit 3 no weight fragment reads, bit 5 no stores.
#ifndef W2X_MLP_EXP
#define W2X_MLP_EXP 0
#endif
// W2X_MLP_PREFETCH 1: a wave requests its NEXT tile's rows after the last first-layer product of the current one, so that they
// travel under the last chunk's GELU, its second-layer products and the epilogue, and the wait at the top of the next tile is
// for loads that are OLDER than the epilogue's stores (vmcnt counts in order: a wait for younger loads also waits for every store
// before them).  Without it a wave loads, waits, computes, stores - and with 12 waves per CU too few bytes are in flight:
// tools/ab/mlp96_variants.sh measured 0.345 ms per launch against 0.312 with the rows served from cache and 0.202 with the
// products and the GELU removed (5 TB/s): the memory phase and the compute phase were adding up.
#ifndef W2X_MLP_PREFETCH
#define W2X_MLP_PREFETCH 1   // round 3 (profiles/r3_mlp96/): 0.3018 ms per launch without, 0.3085 with (request after the last chunk), 0.331 with the
#endif                       // request three chunks earlier: with three waves per SIMD the other two cover a wave's fetch, and the SIMD is busy issuing.
                             // Round 4: on, together with W2X_MLP96_PRIO = 1 - once the chunk loop outranks the row phases a wave's fetch is no longer
                             // covered by its neighbours' row phases, and asking a tile ahead pays (profiles/r4_kernels/mlp96_prio.txt: 0.345 / 0.339 as it was,
                             // 0.324 / 0.324 priority alone, 0.318 / 0.320 with the prefetch, 0.334 / 0.337 prefetch alone)
// W2X_MLP96_PRIO: s_setprio by phase.  1 = the chunk loop (products + GELU) at priority 1, the row phases (fetch, LayerNorm, epilogue, stores) at 0: of the three
// waves of a SIMD the ones that multiply win the issue arbitration over the ones that move rows (-5 % per launch, above).  Other modes measured and not kept: 2 the
// reverse (-1.5 %), 3 only the GELU at 1 (-3 %), 4 as 1 at priority 3 (-3 %), 5 only the products at 1 (+2 %), 6 as 1 with the LayerNorm included (-1 %).
#ifndef W2X_MLP96_PRIO
#define W2X_MLP96_PRIO 1
#endif
#ifndef W2X_MLP_STAGGER
#define W2X_MLP_STAGGER 0   // units of s_sleep 127 (about 8 K cycles) between the start of a SIMD's first, second and third wave; measured +9 % time
#endif
#ifndef W2X_MLP_PFCH
#define W2X_MLP_PFCH 5      // the chunk (0..5) after whose first-layer products the request goes out: earlier = further ahead, 24 registers held longer
#endif

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// Rows move through buffer resources over x / y (32-bit byte offsets, bounds-checked by the hardware): a piece at or beyond the end
// reads zeros and its store is dropped, so the ragged last tile and the prefetch past the last tile need no predicate.  The launcher
// cuts passes of more than kMaxBufBytes into runs.
constexpr size_t kMaxBufBytes = 0xFFF00000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

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
// sums over the two lanes (l, l ^ 32) that hold one token row, for two values at once: the two chains fill each other's permlane
// wait states.  The inputs come straight from v_dot2c chains: a dot result needs 3 wait states before a different VALU may read it,
// and nothing inside an asm statement is padded by the compiler - hence the leading s_nop 2.
__device__ __forceinline__ void halves_sum2(float& a0, float& a1) {
    float b0, b1;
    asm volatile(
        "s_nop 2\n\tv_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\t"
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

constexpr int C = 96, RW = 32, NWV = 12, NTHR = NWV * 64;
constexpr int LDX = C + 8, PPR = C / 8, KS = C / 16, NT = C / 32, NCH = 2 * C / 32;   // 6 k-steps of 16, 3 output tiles of 32 channels, 6 chunks of 32 hidden units
constexpr int NP = RW * PPR / 64;            // flat 16-byte pieces per lane (6)
constexpr int SLAB = RW * LDX * 2;           // bytes per wave
constexpr int W1F = NCH * KS, W2F = NCH * NT * 2;   // KiB fragments of the two matrices (36 + 36)
constexpr int NF = KS;                       // fragments per chunk and matrix = ring registers (6)
static_assert(NF == NT * 2, "one ring serves both products");
constexpr int WBYTES = (W1F + W2F) * 1024;
constexpr int BIAS_OFF = WBYTES + NWV * SLAB;   // b1 [2C] | b2 [C] as fp32: read per chunk / tile through LDS (no vector-memory counter involved)
#ifndef W2X_MLP96_DYN
#define W2X_MLP96_DYN 1      // 1: a workgroup's waves take their tiles from a counter in LDS instead of every twelfth one (round 6: -3.4 % per launch, frame 7.356 / 7.361 / 7.363
                             // against 7.373 / 7.381 / 7.388 / 7.387 ms in alternating runs, profiles/r6_kernels/lib_mlp96_dyn_frame_level.txt; 0: tools/ab/lib_variants.sh "k_mlp96q.hip:-DW2X_MLP96_DYN=0").
                             // The launch that carries the image head keeps the fixed stride: with the counter it spills two registers at 168.
#endif
static_assert(!W2X_MLP96_DYN || W2X_MLP_PREFETCH, "the tile counter is read where the next tile's rows are requested: without the prefetch a wave would stop after its first tile");
constexpr int CTR_OFF = BIAS_OFF + 3 * C * 4;   // (W2X_MLP96_DYN) the workgroup's tile counter
constexpr int SMEM96Q = CTR_OFF + 16;
static_assert(RW * PPR % 64 == 0, "flat piece count");
static_assert(SMEM96Q <= 160 * 1024, "LDS budget");

template <bool TOIMG>
__global__ __launch_bounds__(NTHR, 3) void mlp96q_kernel(const MlpParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
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
        if (tid < 3 * C) ((float*)(smem + BIAS_OFF))[tid] = tid < 2 * C ? p.b1[tid] : p.b2[tid - 2 * C];
        if (tid == 0) *(int*)(smem + CTR_OFF) = 0;
    }
#if W2X_MLP_STAGGER
    // The three waves of a SIMD (wv, wv + 4, wv + 8) leave the barrier together and run the same program: left alone they stay in
    // phase - all in their matrix products, then all in their GELU - and the two pipes take turns.  Start them a third of a tile apart.
    for (int k = 0; k < (wv >> 2) * W2X_MLP_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
#endif
    const float* B1s = (const float*)(smem + BIAS_OFF) + h * 4;
    const float* B2s = B1s + 2 * C;
    __syncthreads();

    // Weight fragments reach the MFMAs through a ring of six registers: a fragment is requested from LDS right after the last MFMA
    // that used its register - W2's during GEMM1 (they land under the GELU), the next chunk's W1's during GEMM2 (they land under the
    // rest of GEMM2) - so no product waits on an LDS round trip.  (Reads placed at their point of use made the compiler wait for
    // each one: 72 exposed LDS latencies per tile.)
    half8 wr[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) wr[i] = *(const half8*)(WL + (size_t)i * 512);
    const int gw = blockIdx.x * NWV + wv, nw = gridDim.x * NWV;
    const unsigned xbytes = (unsigned)(p.M * (C * 2));
    const __amdgpu_buffer_rsrc_t XB = make_rsrc(p.x, xbytes), YB = make_rsrc(p.y, xbytes);
    constexpr unsigned TILE_BYTES = RW * C * 2;
    half8 xr[NP];                                // this tile's rows as flat 16-byte pieces (piece k * 64 + lane)
#define W2X_FETCH(TILE) { const unsigned vo = (unsigned)((W2X_MLP_EXP & 1) ? ((TILE) & 63) : (TILE)) * TILE_BYTES + lane * 16u;   \
        _Pragma("unroll") for (int k = 0; k < NP; ++k) xr[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vo + k * 1024u, 0, W2X_LD_LAST_AUX)); }
    // The workgroup's tiles are the ones its waves would walk with the fixed stride - {(blockIdx.x + j * gridDim.x) * NWV + w} - taken in that order by
    // whichever wave is free: the t-th grab (one ds_add_rtn by lane 0) is tile (blockIdx.x + (t / NWV) * gridDim.x) * NWV + t % NWV, increasing in t, so the
    // first grab past the end ends a wave.  The waves of a SIMD do not progress alike (three share its issue slots); with fixed strides the slowest wave's
    // last tiles run beside empty slots (10.0 of 12 waves resident on average, profiles/r5_final/pmc_sq.summary.txt).  Which wave computes a tile does not
    // change a bit of it.
    constexpr bool DYN = W2X_MLP96_DYN && !TOIMG;
    auto grab = [&]() -> int {
        int t = 0;
        if (lane == 0) t = atomicAdd((int*)(smem + CTR_OFF), 1);
        t = __builtin_amdgcn_readfirstlane(t);
        const int tl = (blockIdx.x + (t / NWV) * (int)gridDim.x) * NWV + t % NWV;
        return tl < ntiles ? tl : ntiles;            // (ntiles: one tile past the end - its rows read zeros, its stores are dropped, the loop ends)
    };
    int tile = gw, next_tile = ntiles;
    if constexpr (DYN) tile = grab();
    if (tile < ntiles) W2X_FETCH(tile)
#if W2X_MLP_PREFETCH
    // NP stores that the hardware drops (offset past the end), so that the first pass through the loop sees the same queue as every
    // later one - the tile's loads followed by NP stores - and the compiler's wait at the top of the loop is "all but the NP youngest"
    // instead of "everything" (its counter analysis merges the two ways into the loop and keeps the stricter wait)
#pragma unroll
    for (int k = 0; k < NP; ++k) __builtin_amdgcn_raw_buffer_store_b128(uint4v{}, YB, 0xFFFFF000u + k * 16u, 0, 0);   // (distinct offsets: identical stores would be merged)
#endif
#pragma unroll 1
    for (; tile < ntiles; tile = DYN ? next_tile : tile + nw) {
        const long row0 = (long)((W2X_MLP_EXP & 1) ? (tile & 63) : tile) * RW;
        const long nrows = p.M - row0 < RW ? p.M - row0 : RW;
        // ---- x rows: flat coalesced pieces -> slab (the raw rows stay there for the residual add)
#if !W2X_MLP_PREFETCH
        W2X_FETCH(tile)
#endif
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            *(half8*)(Xw + r * LDX + c * 8) = xr[k];
        }
        W2X_PHASE_FENCE();
#if defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 6
        __builtin_amdgcn_s_setprio(1);
#endif
        // ---- LayerNorm in fragment layout: lane (r32, h) holds channels ks*16 + 8h .. +7 of row r32, so the row sums are the
        //      lane's own KS pieces plus those of lane ^ 32; the normalised pieces are the B operand registers of GEMM1
        half8 xreg[KS];
        {
            half8 raw[KS];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { raw[ks] = *(const half8*)(Xw + r32 * LDX + ks * 16 + h * 8); sum_sq8(raw[ks], s, q); }
            halves_sum2(s, q);
            const float mean = s * (1.f / C);
            const float rstd = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps);
            const float nm = -mean * rstd;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xreg[ks] = norm8(raw[ks], rstd, nm);
        }
        // GEMM2 accumulators (rows = output channels 32nt + 8q + 4h + j in register 4q + j, columns = tokens) start from b2
        float16v acc2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v b = *(const float4v*)(B2s + nt * 32 + q * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[nt][4 * q + j] = b[j];
            }
#if defined(W2X_MLP96_PRIO) && (W2X_MLP96_PRIO == 1 || W2X_MLP96_PRIO == 5)   // s_setprio by phase: 1 = the chunk loop at priority 1, row phases at 0; 2 = the reverse; 3 = only the GELU at 1;
        __builtin_amdgcn_s_setprio(1);                                          // 4 = as 1 at priority 3; 5 = only the matrix products at 1 (GELU at 0); 6 = as 1, LayerNorm included
#elif defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#elif defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 4
        __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            // GEMM1 (transposed): acc1 = W1[32ch ..][:] * Xn^T   (rows = hidden units of the chunk, columns = tokens), from b1
            float16v acc1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v b = *(const float4v*)(B1s + ch * 32 + q * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[4 * q + j] = b[j];
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (!(W2X_MLP_EXP & 4)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[i], xreg[i], acc1, 0, 0, 0);
                else acc1[i] += (float)xreg[i][0] + (float)wr[i][0];
                if (!(W2X_MLP_EXP & 8)) wr[i] = *(const half8*)(WL + (size_t)(W1F + ch * NF + i) * 512);         // ring: W2 fragment i of this chunk, in flight under the GELU
                W2X_RING_FENCE();
            }
#if W2X_MLP_PREFETCH
            if (ch == W2X_MLP_PFCH) {                                           // the next tile's rows (past the last tile: zeros)
                if constexpr (DYN) { next_tile = grab(); W2X_FETCH(next_tile) } else { W2X_FETCH(tile + nw) }
                W2X_RING_FENCE();
            }
#endif
            // GELU in place; registers 8s .. 8s+7 of a lane are hidden rows 16s + 8(j >> 2) + 4h + (j & 3) of its token column ->
            // B fragment of k-step s of GEMM2 in the k order W2 is stored in (fragorder.h frag32_w2)
            half8 a2[2];
#if defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 3
            __builtin_amdgcn_s_setprio(1);
#elif defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 5
            __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (W2X_MLP_EXP & 2) {
                    a2[s2] = (half8){(_Float16)acc1[8 * s2], (_Float16)acc1[8 * s2 + 1], (_Float16)acc1[8 * s2 + 2], (_Float16)acc1[8 * s2 + 3],
                                     (_Float16)acc1[8 * s2 + 4], (_Float16)acc1[8 * s2 + 5], (_Float16)acc1[8 * s2 + 6], (_Float16)acc1[8 * s2 + 7]};
                    continue;
                }
                const float2v g0 = gelu_fast2((float2v){acc1[8 * s2 + 0], acc1[8 * s2 + 1]});
                const float2v g1 = gelu_fast2((float2v){acc1[8 * s2 + 2], acc1[8 * s2 + 3]});
                const float2v g2 = gelu_fast2((float2v){acc1[8 * s2 + 4], acc1[8 * s2 + 5]});
                const float2v g3 = gelu_fast2((float2v){acc1[8 * s2 + 6], acc1[8 * s2 + 7]});
                a2[s2] = (half8){(_Float16)g0[0], (_Float16)g0[1], (_Float16)g1[0], (_Float16)g1[1],
                                 (_Float16)g2[0], (_Float16)g2[1], (_Float16)g3[0], (_Float16)g3[1]};
            }
#if defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 3
            __builtin_amdgcn_s_setprio(0);
#elif defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 5
            __builtin_amdgcn_s_setprio(1);
#endif
            // GEMM2 (transposed): acc2[nt] += W2[32nt ..][chunk] * H[tokens][chunk]^T, fragment i = (nt, k-step) = (i >> 1, i & 1)
            const int nch = ch + 1 < NCH ? ch + 1 : 0;          // after the last chunk: the first chunk's fragments for the next tile
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (!(W2X_MLP_EXP & 4)) acc2[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[i], a2[i & 1], acc2[i >> 1], 0, 0, 0);
                else acc2[i >> 1][i] += (float)a2[i & 1][0] + (float)wr[i][1];
                if (!(W2X_MLP_EXP & 8)) wr[i] = *(const half8*)(WL + (size_t)(nch * NF + i) * 512);               // ring: W1 fragment i (k-step i) of the next chunk
                W2X_RING_FENCE();
            }
        }
#if defined(W2X_MLP96_PRIO) && (W2X_MLP96_PRIO == 1 || W2X_MLP96_PRIO >= 4)
        __builtin_amdgcn_s_setprio(0);
#elif defined(W2X_MLP96_PRIO) && W2X_MLP96_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        W2X_PHASE_FENCE();
        half8 wfh[2][3];                          // TOIMG: the head's first six weight fragments, requested here so that they land under the epilogue below
        if (TOIMG) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) wfh[t][ks] = *(const half8*)((const _Float16*)p.ti_w + lane * 8 + (size_t)(t * 3 + ks) * 512);
            W2X_RING_FENCE();
        }
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
            for (int q = 0; q < 4; ++q)
                *(half4*)(Xw + r32 * LDX + nt * 32 + q * 8 + h * 4) = (half4){(_Float16)acc2[nt][4 * q], (_Float16)acc2[nt][4 * q + 1], (_Float16)acc2[nt][4 * q + 2], (_Float16)acc2[nt][4 * q + 3]};
        W2X_PHASE_FENCE();
        const unsigned yo = (unsigned)row0 * (C * 2) + lane * 16u;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            const half8 o = *(const half8*)(Xw + r * LDX + c * 8) + xres[k];     // fp16 + fp16 rounded once == fp32 add rounded to fp16
            if (TOIMG) *(half8*)(Xw + r * LDX + c * 8) = o;                      // the rows go through the image head below instead of to y
            else if (!(W2X_MLP_EXP & 32)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), YB, yo + k * 1024u, 0, W2X_ST_AUX);
        }
        if (TOIMG) {
            // ---- image head: Linear 96 -> 64 = 4x4 sub-pixels x 4 stored channels, Clip, DepthToSpace(4) - the sums of toimage_kernel (k_pixgemm.hip), transposed:
            // out^T = W y^T (rows = output columns, columns = tokens), so lane (token fr16, g4) of n-tile t ends with columns 16 t + 4 g4 .. + 3 = the four channels
            // of sub-pixel (dy, dx) = (t, g4) of its token - a finished pixel, stored as 8 bytes; the four lane groups of a token write 32 contiguous bytes, the 16
            // tokens of an m-tile 512.  No trip through LDS for the result.  The head's 12 weight fragments come from L2 per tile (12 KiB: LDS is full, and their
            // 24 registers at a time are only free here, where the accumulators are dead).
            W2X_PHASE_FENCE();
            const int fr16 = lane & 15, g4 = lane >> 4;
            const _Float16* Wt = (const _Float16*)p.ti_w + lane * 8;
            // A tile lies inside one image (rows per image are a multiple of 32: mlp96q_supported) and spans at most two rows of its token map (row width >= 32)
            const int row0i = __builtin_amdgcn_readfirstlane((int)(row0 + p.ti_row0));       // row index in the whole pass (launch_mlp96q cuts passes beyond 4 GB into runs)
            const int bimg = row0i / p.ti_Mrows, ml0 = row0i - bimg * p.ti_Mrows;
            const int oy0 = ml0 / p.ti_aW, ox0 = ml0 - oy0 * p.ti_aW;
            _Float16* __restrict__ Og = (_Float16*)p.ti_out + (size_t)bimg * p.ti_Hs * p.ti_Ws * 4;
            const half2v lo2 = {(_Float16)p.ti_lo, (_Float16)p.ti_lo}, hi2 = {(_Float16)p.ti_hi, (_Float16)p.ti_hi};
#pragma unroll
            for (int th = 0; th < 2; ++th) {      // two n-tiles at a time
                half8 wf[2][3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) wf[t][ks] = th == 0 ? wfh[t][ks] : *(const half8*)(Wt + (size_t)((2 + t) * 3 + ks) * 512);
                float4v acc[2][2];                // [n-tile of the pair][m-tile]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4v b = *(const float4v*)(p.ti_b + (2 * th + t) * 16 + g4 * 4);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) acc[t][tt] = b;
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        const half8 xa = *(const half8*)(Xw + (tt * 16 + fr16) * LDX + ks * 32 + g4 * 8);
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][ks], xa, acc[t][tt], 0, 0, 0);
                    }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    int ox = ox0 + tt * 16 + fr16, oy = oy0;
                    if (ox >= p.ti_aW) { ox -= p.ti_aW; ++oy; }
                    const bool ok = row0 + tt * 16 + fr16 < p.M;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        half2v v01 = {(_Float16)acc[t][tt][0], (_Float16)acc[t][tt][1]}, v23 = {(_Float16)acc[t][tt][2], (_Float16)acc[t][tt][3]};
                        if (p.ti_clip) {          // bounds that fp16 represents exactly (mlp96q_supported): clamping the rounded value in fp16 is clamping it in fp32 and rounding again
                            // max, then min, as instructions: this file is built with -fno-honor-nans, under which the compiler may reorder the pair (equal for
                            // numbers, not for a NaN: v_pk_max / v_pk_min return the other operand, so max-then-min sends a NaN to clip_lo like toimage_kernel's
                            // fminf(fmaxf(v, lo), hi), min-then-max would send it to clip_hi)
                            asm("v_pk_max_f16 %0, %1, %2\n\tv_pk_min_f16 %0, %0, %3" : "=&v"(v01) : "v"(v01), "v"(lo2), "v"(hi2));
                            asm("v_pk_max_f16 %0, %1, %2\n\tv_pk_min_f16 %0, %0, %3" : "=&v"(v23) : "v"(v23), "v"(lo2), "v"(hi2));
                        }
                        const half4 px = {v01[0], v01[1], v23[0], v23[1]};
                        if (ok) *(half4*)(Og + ((size_t)(oy * 4 + 2 * th + t) * p.ti_Ws + ox * 4 + g4) * 4) = px;
                    }
                }
            }
        }
        if (!TOIMG && p.stats_out) {   // an un-fused consumer wants the LayerNorm statistics of the produced rows: put them back into the slab
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
                *(half8*)(Xw + r * LDX + c * 8) = *(const half8*)(Xw + r * LDX + c * 8) + xres[k];
            }
        }
        W2X_PHASE_FENCE();
        if (!TOIMG && p.stats_out && lane < nrows) {   // LayerNorm statistics of the produced rows for an un-fused consumer
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int c = 0; c < PPR; ++c) sum_sq8(*(const half8*)(Xw + lane * LDX + c * 8), s, q);
            const float mean = s * (1.f / C);
            p.stats_out[2 * (row0 + lane)] = mean;
            p.stats_out[2 * (row0 + lane) + 1] = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out);
        }
        W2X_PHASE_FENCE();   // the next tile's rows overwrite the slab
    }
#undef W2X_FETCH
}

}  // namespace

bool mlp96q_supported(const MlpParams& p) { return p.C == C && p.w1_frag && p.w2_frag && p.frag32 && (!p.ti_w || (p.ti_b && p.ti_out && !p.stats_out && p.ti_Mrows > 0 && p.ti_Mrows % RW == 0 && p.ti_aW >= RW && p.M < 0x7FFFFFFFl &&
                                                                 (!p.ti_clip || ((float)(_Float16)p.ti_lo == p.ti_lo && (float)(_Float16)p.ti_hi == p.ti_hi)))); }

hipError_t launch_mlp96q(const MlpParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    static unsigned lds_ok_t = 0;
    if (hipError_t e = p.ti_w ? ensure_dynamic_lds((const void*)mlp96q_kernel<true>, SMEM96Q, lds_ok_t) : ensure_dynamic_lds((const void*)mlp96q_kernel<false>, SMEM96Q, lds_ok); e != hipSuccess) return e;
    static int cus[32] = {0};     // compute units per device (one resident workgroup each)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    int ncu = __atomic_load_n(&cus[dev & 31], __ATOMIC_RELAXED);
    if (ncu == 0) {
        if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        __atomic_store_n(&cus[dev & 31], ncu, __ATOMIC_RELAXED);
    }
    // the kernel addresses x / y with 32-bit byte offsets (and prefetches one tile stride past its last tile): longer passes run in pieces
    const long max_rows = (long)((kMaxBufBytes / (C * 2)) / RW) * RW;
    for (long r0 = 0; r0 < p.M; r0 += max_rows) {
        MlpParams q = p;
        q.M = std::min(max_rows, p.M - r0);
        q.x = (const char*)p.x + (size_t)r0 * C * 2; q.y = (char*)p.y + (size_t)r0 * C * 2;
        if (p.stats_out) q.stats_out = p.stats_out + 2 * r0;
        const long ntiles = (q.M + RW - 1) / RW;
        const int grid = (int)std::min<long>((ntiles + NWV - 1) / NWV, ncu);
        if ((ntiles + (long)grid * NWV) * (long)(RW * C * 2) > 0xFFFFFFFFl) return hipErrorInvalidValue;   // (cannot happen below kMaxBufBytes with <= 1024 CUs)
        if (p.ti_w) {
            q.ti_row0 = r0;
            hipLaunchKernelGGL(mlp96q_kernel<true>, dim3(grid), dim3(NTHR), SMEM96Q, s, q, (int)ntiles);
        } else hipLaunchKernelGGL(mlp96q_kernel<false>, dim3(grid), dim3(NTHR), SMEM96Q, s, q, (int)ntiles);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace w2x
