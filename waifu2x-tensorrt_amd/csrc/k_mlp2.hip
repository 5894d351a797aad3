// Fused transformer MLP branch for gfx950, wave-private rows:   y = x + W2 * gelu(W1 * LayerNorm(x) + b1) + b2
// Every wave owns 16*TT token rows from the first load to the last store:
//   * x rows arrive as one flat, fully coalesced stream into the wave's LDS slab; each lane then reads the 16-byte pieces it
//     uses as MFMA fragments, LayerNorm statistics are the lane's own sums plus two row swaps, and the normalised pieces are
//     the operand registers of GEMM1 as they stand (LDS program order is the only ordering needed inside a wave, see
//     W2X_PHASE_FENCE);
//   * weights: the four waves of a workgroup stage each 32-hidden-unit chunk once into LDS by LDS-DMA (global_load_lds: the
//     fragment-major copy of engine.cpp makes a fragment one contiguous KiB = one wave instruction, no registers in between),
//     double-buffered, one workgroup barrier per chunk - per-wave weight streams from L2 were 56 % of the C = 192 kernel's time -
//     and read them into a ring of six registers, each fragment six fragments ahead of its use;
//   * GEMM1 is computed transposed (rows = 32 hidden units of the chunk, columns = tokens) so that the GELU'd accumulators
//     are, as they stand, the B fragments of GEMM2 with a permuted k order (W2 is stored with the same permutation) - no LDS
//     round trip between the two products; GEMM2 is transposed too (rows = output channels), so a lane ends with 4 consecutive
//     channels of a token: b2 is the initial accumulator and the tile reaches the slab in 8-byte stores;
//   * the result tile goes through the slab once so that residual add and stores are flat 16-byte pieces again; at C = 96 the
//     raw x rows are still in the slab at that point (the weight buffers sit behind the slabs), so x is read from HBM once.
// TT = 2 for both widths, which leaves C = 96 at 3 waves per SIMD.
#include "kernels.h"
#ifndef W2X_GELU_DEG
#define W2X_GELU_DEG 4   // coefficients of q(u): 6 -> 3.1e-7, 5 -> 7.1e-7, 4 -> 8.7e-6 absolute error of GELU (tools/fit_gelu.py).  4: a third of
                         // the fp16 rounding of the smallest hidden values that matter, network parity unchanged (2.0 ULP16 on every full-width
                         // graph, same mean error), MLP kernels 5-7 % faster (round 2, profiles/r2_final/gelu_degree_ab.txt; now: tools/ab/lib_variants.sh "k_mlp2.hip:-DW2X_GELU_DEG=6")
#endif

#include <cstdlib>

// C = 192 geometry (tools/ab/mlp192_variants.sh times the alternatives): token tiles of 16 rows per wave, waves per workgroup, and
// the workgroups per CU the register budget is set for
#ifndef W2X_MLP192_TT
#define W2X_MLP192_TT 2
#endif
#ifndef W2X_MLP192_NW
#define W2X_MLP192_NW 4
#endif
#ifndef W2X_MLP192_WPS
#define W2X_MLP192_WPS 2
#endif
#ifndef W2X_MLP192_KEEP
#define W2X_MLP192_KEEP 0     // 1: weight buffers behind the slabs, the raw rows stay in LDS for the residual add (no second fetch); needs 8 waves per
#endif                        // workgroup to fit the CU (one workgroup of 150 KB instead of two of 51 KB)

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// GELU(x) = max(x,0) - 0.5 u 2^-q(u), u = min(|x|, 6.5): tools/fit_gelu.py (|err| < 8.7e-6 with the four-coefficient q, W2X_GELU_DEG)
// Two values at a time: the polynomial, the products and the final fma are v_pk_*_f32 (one issue slot for both values);
// min / max / exp2 have no packed form.  Same operations per element as the scalar form, so the results are identical.
typedef float float2v __attribute__((ext_vector_type(2)));
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
// sum over the four 16-lane rows of a wave (see k_swinattn.hip for why the swaps are inline asm on two registers)
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_sum(float v) { float a = v, b = v; swap16(a, b); v = a + b; a = v; b = v; swap32(a, b); return a + b; }
// (x * rstd + nm) on 8 halves with fp32 arithmetic: v_fma_mixlo / mixhi read the f16 halves directly and write f16 (one instruction
// per element; the compiler's own lowering converts both ways around a packed fp32 fma)
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
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

// The phases below hand data from lane to lane through the wave's own LDS slab.  The hardware executes a wave's LDS
// instructions in order, so no s_barrier / s_waitcnt is needed; the compiler-level fence (no instruction emitted) keeps
// hipcc from forwarding a lane's own store to its later load or moving slab accesses across a phase boundary.
// History: a first version exchanged per-row LayerNorm statistics through a small float table in the slab
// (ds_write_b64 by lane = row, ds_read_b64 by the fragment lanes a few instructions later).  That exchange returned stale
// values on the second wave of a SIMD at full problem size (tools/ab/mlp_ab.hip reproduces it: first workgroup per CU always
// right, co-resident ones wrong, not cured by s_waitcnt / s_barrier) and was replaced by register swaps.
#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")
#define W2X_RING_FENCE() asm volatile("" ::: "memory")   // keeps a ring refill where it is written (the scheduler would sink it to its use)

// Timing experiments on mlp2q_kernel (results are wrong; tools/ab/mlp192_variants.sh): bit 0 no barrier per chunk, bit 1 no weight staging after
// chunk 0 (and no wait for it), bit 2 GELU replaced by the bare conversion, bit 3 no matrix products, bit 4 every wave fetches its rows (and the residual
// rows) from the first 12 KiB of x (cache hits: no HBM latency or bandwidth on the way in), bit 5 no stores.
#ifndef W2X_MLP2Q_PIPE
#define W2X_MLP2Q_PIPE 0     // 1: the chunk loop software-pipelined (first-layer products of chunk c + 1 between the pieces of chunk c's GELU), see mlp2q_kernel
#endif
#ifndef W2X_MLP2Q_EXP
#define W2X_MLP2Q_EXP 0
#endif
// 1: the first chunk's weights get a buffer of their own behind the slabs (78 KB per workgroup, still two per CU) and are staged when the workgroup starts,
// under the row fetch and the LayerNorm; 0: both buffers alias the slabs, chunk 0 is staged after the rows are in registers and waited for on the spot.
// Measured equal (0.3025 / 0.0794 against 0.3004 / 0.0790 ms, profiles/r3_kernels/mlp2q_early0.txt): the CU's other workgroup already covers the wait.  Off.
#ifndef W2X_MLP2Q_EARLY0
#define W2X_MLP2Q_EARLY0 0
#endif
#ifndef W2X_MLP2Q_SPLITACC
#define W2X_MLP2Q_SPLITACC 0
#endif
template <int C, int TT, int NW>
struct Mlp2Cfg {
    static constexpr int RW = 16 * TT;           // rows per wave
    static constexpr int NWV = NW;               // waves per workgroup
    static constexpr int BM = NWV * RW;          // rows per workgroup
    static constexpr int LDX = C + 8;            // slab row stride in halves: 16-byte pieces rotate over the banks
    static constexpr int PPR = C / 8;            // 16-byte pieces per row
    static constexpr int KS = C / 32;            // k-steps of GEMM1
    static constexpr int NT = C / 16;            // output n-tiles of GEMM2
    static constexpr int NCH = 2 * C / 32;       // hidden chunks of 32
    static constexpr int NP = RW * PPR / 64;     // flat 16-byte pieces per lane
    static constexpr int SLAB = RW * LDX * 2;    // bytes per wave
    static constexpr int NF = 2 * KS + NT;       // weight fragments (KiB) per hidden chunk
    static constexpr int WBUF = NF * 1024;       // one staged chunk, two buffers
    static constexpr int RING = 6;               // weight-fragment registers of a wave
    // KEEP (C = 96): the weight buffers sit behind the slabs (50.6 KB per workgroup, still 3 workgroups per CU), so the raw x rows stay
    // in the slab and serve the residual add - x is read from HBM once.  Otherwise the buffers alias the slabs (x lives in
    // registers by then) and the residual rows are fetched a second time.
    static constexpr bool KEEP = C == 96 || (W2X_MLP192_KEEP && NW != 4);   // (the 32x32x16 kernel below is the <192, 2, 4> geometry and aliases its buffers)
    static constexpr int WORK = KEEP ? NWV * SLAB + 2 * WBUF : (NWV * SLAB > 2 * WBUF ? NWV * SLAB : 2 * WBUF);
    // b1 [2C] | b2 [C] as fp32 behind the work area: the per-chunk bias reads are LDS reads.  As global loads they shared the vector
    // memory counter with the LDS-DMA staging of the NEXT chunk, and the wait for a chunk's two bias vectors (s_waitcnt vmcnt(0), in
    // order) was a wait for that staging: the double buffering did not overlap anything.
    static constexpr int BIAS_OFF = WORK;
    static constexpr int SMEM = WORK + 3 * C * 4;
    static_assert(NF % NWV == 0 && NF % RING == 0 && 2 * KS >= RING, "fragments per wave / ring slots");
    static_assert(RW * PPR % 64 == 0, "flat piece count");
};

template <int C, int TT, int NW>
// (the second launch bound is hipcc's minimum number of waves per SIMD, not blocks per CU)
__global__ __launch_bounds__(NW * 64, (C == 96 ? 3 : W2X_MLP192_WPS) * NW / 4) void mlp2_kernel(const MlpParams p) {
    using K = Mlp2Cfg<C, TT, NW>;
    constexpr int RW = K::RW, LDX = K::LDX, PPR = K::PPR, KS = K::KS, NT = K::NT, NCH = K::NCH, NP = K::NP, NF = K::NF, RING = K::RING;
    constexpr int NFW = NF / K::NWV;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* Xw = (_Float16*)(smem + wv * K::SLAB);          // [RW][LDX]
    unsigned char* const WBb = smem + (K::KEEP ? K::NWV * K::SLAB : 0);   // two weight buffers of NF fragments [64 lanes][8]

    const long row0 = ((long)blockIdx.x * K::NWV + wv) * RW;       // first row of this wave
    const long nrows = p.M - row0 < RW ? p.M - row0 : RW;     // may be <= 0: the wave then only runs dead arithmetic
    // rows through buffer resources (32-bit byte offsets, bounds-checked: pieces past the last row read zeros, their stores are dropped;
    // the launcher cuts passes of more than kMaxBufBytes into runs)
    const unsigned xbytes = (unsigned)(p.M * (C * 2));
    const __amdgpu_buffer_rsrc_t XB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t YB = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, xbytes, 0x00020000);
    const unsigned vo = nrows > 0 ? (unsigned)row0 * (C * 2) + lane * 16u : 0xFFFFC000u;   // (no rows: every piece past the end, without wrapping)
    const _Float16* __restrict__ W1 = (const _Float16*)p.w1_frag + lane * 8;   // [NCH*2 row tiles][KS][64][8]
    const _Float16* __restrict__ W2 = (const _Float16*)p.w2_frag + lane * 8;   // [NCH][NT][64][8], k order of the GELU'd accumulators

    // ---- weights: the four waves stage each 32-hidden-unit chunk once into LDS (fragment-major: a fragment is one contiguous KiB,
    //      which is exactly what one LDS-DMA instruction of a wave moves - global_load_lds, 16 bytes per lane, no registers, no
    //      ds_write), two buffers, wave w brings fragments w*NFW .. of the next chunk while the current one is consumed.
    auto frag_src = [&](int ch, int f) { return f < 2 * KS ? W1 + (size_t)(ch * 2 * KS + f) * 512 : W2 + (size_t)(ch * NT + (f - 2 * KS)) * 512; };
    auto stage = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int f = wv * NFW + i;
            __builtin_amdgcn_global_load_lds((const void*)frag_src(ch, f), (__attribute__((address_space(3))) void*)(WBb + (size_t)(ch & 1) * K::WBUF + (size_t)f * 1024), 16, 0, 0);
        }
    };
    // fragment j of a chunk in the order the products consume them: GEMM1 (k-step j / 2, hidden half j & 1), then GEMM2 (n-tile j - 2 KS)
    auto lds_frag = [&](int ch, int j) {
        const int f = j < 2 * KS ? (j & 1) * KS + (j >> 1) : j;
        return *(const half8*)(WBb + (size_t)(ch & 1) * K::WBUF + (size_t)f * 1024 + lane * 16);
    };
    if (K::KEEP) stage(0);                     // separate buffers: under the row loads and the LayerNorm
    for (int i = tid; i < 3 * C; i += K::NWV * 64) ((float*)(smem + K::BIAS_OFF))[i] = i < 2 * C ? p.b1[i] : p.b2[i - 2 * C];   // (first read after two barriers)
    const float* B1s = (const float*)(smem + K::BIAS_OFF) + g * 4;
    const float* B2s = B1s + 2 * C;

    // ---- x rows: flat coalesced load -> slab
    {
        half8 xr[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) xr[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vo + k * 1024u, 0, 0));
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            *(half8*)(Xw + r * LDX + c * 8) = xr[k];
        }
    }
    W2X_PHASE_FENCE();
    // ---- LayerNorm in fragment layout: lane (fr, g) holds channels ks*32 + 8g .. +7 of row 16tt + fr, so the row sums are the
    //      lane's own KS pieces plus the three other lane groups (two row swaps); the normalised pieces are the operand registers
    //      of GEMM1 as they stand (rows without data hold zeros: 0 * rstd - 0)
    half8 xreg[TT][KS];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        half8 raw[KS];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { raw[ks] = *(const half8*)(Xw + (tt * 16 + fr) * LDX + ks * 32 + g * 8); sum_sq8(raw[ks], s, q); }
        s = rows_sum(s);
        q = rows_sum(q);
        const float mean = s * (1.f / C);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps);   // the argument is >= eps
        const float nm = -mean * rstd;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xreg[tt][ks] = norm8(raw[ks], rstd, nm);
    }
    W2X_PHASE_FENCE();
    if (!K::KEEP) {
        __syncthreads();                       // every wave holds its rows in registers: the slab area becomes weight buffers
        stage(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's share of chunk 0 has landed
    __syncthreads();

    // Weight fragments reach the MFMAs through a ring of six registers: a fragment is requested from LDS right after the last MFMA
    // that used its register, six fragments (12 MFMAs) ahead of its use, so no product waits on an LDS round trip.  (Reads placed at
    // their point of use made the compiler wait for each one.)
    half8 wr[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) wr[i] = lds_frag(0, i);

    // GEMM2 is computed transposed as well (rows = output channels, columns = tokens): a lane ends with 4 consecutive channels of
    // one token (8-byte slab stores) and b2 is the initial accumulator
    float4v acc2[TT][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float4v b2v = *(const float4v*)(B2s + j * 16);
#pragma unroll
        for (int i = 0; i < TT; ++i) acc2[i][j] = b2v;
    }

    half8 xres[NP];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {   // fully unrolled: the ring registers are renamed statically
        if (ch + 1 < NCH) stage(ch + 1);       // into the other buffer (last read before the previous barrier); in flight under this chunk
        // GEMM1 (transposed): acc1[ht][tt] = W1[32ch + 16ht ..][:] * Xn[16tt ..][:]^T   (rows = hidden, columns = tokens), from b1
        float4v acc1[2][TT];
        {
            const float4v be = *(const float4v*)(B1s + ch * 32);
            const float4v bo = *(const float4v*)(B1s + ch * 32 + 16);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) { acc1[0][tt] = be; acc1[1][tt] = bo; }
        }
#pragma unroll
        for (int j = 0; j < 2 * KS; ++j) {
            const int ks = j >> 1, ht = j & 1;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc1[ht][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[j % RING], xreg[tt][ks], acc1[ht][tt], 0, 0, 0);
            wr[j % RING] = lds_frag(ch, j + RING);
            W2X_RING_FENCE();
        }
        if (!K::KEEP && ch == NCH - 1) {   // the residual rows, requested as soon as the normalised copies have served their last product:
#pragma unroll                           // they travel under the last GELU and second-layer products (round 2 fetched them in the epilogue and waited)
            for (int k = 0; k < NP; ++k) xres[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vo + k * 1024u, 0, W2X_LD_LAST_AUX));
        }
        // GELU in place; lane holds hidden rows 16ht + 4g + j of token column fr -> B fragment of GEMM2 for the
        // k order (ht 0: slots 0..3, ht 1: slots 4..7)
        half8 a2[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const float4v e = acc1[0][tt], o = acc1[1][tt];
            const float2v g0 = gelu_fast2((float2v){e[0], e[1]});
            const float2v g1 = gelu_fast2((float2v){e[2], e[3]});
            const float2v g2 = gelu_fast2((float2v){o[0], o[1]});
            const float2v g3 = gelu_fast2((float2v){o[2], o[3]});
            a2[tt] = (half8){(_Float16)g0[0], (_Float16)g0[1], (_Float16)g1[0], (_Float16)g1[1],
                             (_Float16)g2[0], (_Float16)g2[1], (_Float16)g3[0], (_Float16)g3[1]};
        }
        // GEMM2 (transposed): acc2[tt][nt] += W2[16nt ..][chunk] * H[tokens][chunk]^T
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int j = 2 * KS + nt;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc2[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[j % RING], a2[tt], acc2[tt][nt], 0, 0, 0);
            if (j + RING < NF) { wr[j % RING] = lds_frag(ch, j + RING); W2X_RING_FENCE(); }
        }
        if (ch + 1 < NCH) __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): this wave's share of the next chunk has landed
        __syncthreads();                       // (after the last chunk: every wave is done with the weight buffers the slabs alias)
        if (ch + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < RING; ++i) wr[i] = lds_frag(ch + 1, i);
        }
    }

    W2X_PHASE_FENCE();
    // ---- epilogue: residual pieces first (KEEP: from the slab, which still holds the raw rows; otherwise a second fetch),
    //      accumulators -> fp16 tile in the slab, then flat pieces
    if (K::KEEP) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            xres[k] = *(const half8*)(Xw + r * LDX + c * 8);
        }
    }
    W2X_PHASE_FENCE();
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
            *(half4*)(Xw + (tt * 16 + fr) * LDX + nt * 16 + g * 4) = (half4){(_Float16)acc2[tt][nt][0], (_Float16)acc2[tt][nt][1], (_Float16)acc2[tt][nt][2], (_Float16)acc2[tt][nt][3]};
    W2X_PHASE_FENCE();
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
        const half8 o = *(const half8*)(Xw + r * LDX + c * 8) + xres[k];     // fp16 + fp16 rounded once == fp32 add rounded to fp16
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), YB, vo + k * 1024u, 0, W2X_ST_AUX);
        if (p.stats_out) *(half8*)(Xw + r * LDX + c * 8) = o;
    }
    W2X_PHASE_FENCE();
    if (p.stats_out && lane < nrows) {   // LayerNorm statistics of the produced rows for an un-fused consumer
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int c = 0; c < PPR; ++c) sum_sq8(*(const half8*)(Xw + lane * LDX + c * 8), s, q);
        const float mean = s * (1.f / C);
        p.stats_out[2 * (row0 + lane)] = mean;
        p.stats_out[2 * (row0 + lane) + 1] = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out);
    }
}

// ---- the same schedule on v_mfma_f32_32x32x16_f16 (round 3; C = 192 in the engine).  Why: k_mlp96q.hip.  A wave still owns 32 token rows; a chunk is
// 32 hidden units = ONE 32 x 32 accumulator of the transposed first product (12 k-steps of 16 channels), whose registers 8s .. 8s+7 are the B fragment of
// k-step s of the second product as they stand (W2 stored in that k order: fragorder.h frag32_w2); the second product is 6 tiles of 32 output channels
// x 2 k-steps.  Same FLOP, same 24 KiB of fragments per chunk, half the matrix instructions, LayerNorm sums with one lane swap.
typedef float float16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void halves_sum2(float& a0, float& a1) {   // sums over the lanes l, l ^ 32 of two values at once (k_mlp96q.hip)
    float b0, b1;
    asm volatile(
        "s_nop 2\n\tv_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
        "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3"
        : "+v"(a0), "+v"(a1), "=&v"(b0), "=&v"(b1));
}

template <int C, int NW>
__global__ __launch_bounds__(NW * 64, 2 * NW / 4) void mlp2q_kernel(const MlpParams p) {
    using K = Mlp2Cfg<C, 2, NW>;
    constexpr int RW = K::RW, LDX = K::LDX, PPR = K::PPR, NP = K::NP, RING = K::RING;
    constexpr int KS = C / 16, NT = C / 32, NCH = 2 * C / 32, NF = KS + 2 * NT, NFW = NF / K::NWV;
    static_assert(NF == K::NF && NF % K::NWV == 0 && NF % RING == 0 && KS >= RING && !K::KEEP, "fragments per chunk");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    _Float16* Xw = (_Float16*)(smem + wv * K::SLAB);          // [RW][LDX]
    // two weight buffers of NF fragments [64 lanes][8]: odd chunks in the slab area (the rows are in registers by then), even chunks behind it (EARLY0) or there as well
    constexpr int BUF0 = W2X_MLP2Q_EARLY0 ? K::NWV * K::SLAB : 0, BUF1 = W2X_MLP2Q_EARLY0 ? 0 : K::WBUF;
    constexpr int BIASQ = W2X_MLP2Q_EARLY0 ? K::NWV * K::SLAB + K::WBUF : K::BIAS_OFF;
    auto wbuf = [&](int ch) { return smem + ((ch & 1) ? BUF1 : BUF0); };

    const long row0 = ((long)blockIdx.x * K::NWV + wv) * RW;
    const long nrows = p.M - row0 < RW ? p.M - row0 : RW;
    const unsigned xbytes = (unsigned)(p.M * (C * 2));
    const __amdgpu_buffer_rsrc_t XB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t YB = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, xbytes, 0x00020000);
    const unsigned vo = nrows > 0 ? (unsigned)row0 * (C * 2) + lane * 16u : 0xFFFFC000u;
    const unsigned vl = (W2X_MLP2Q_EXP & 16) ? lane * 16u : vo;     // (timing experiment: where the rows are read from)
    const _Float16* __restrict__ W1 = (const _Float16*)p.w1_frag + lane * 8;   // frag32_major: [NCH row tiles of 32][KS][64][8]
    const _Float16* __restrict__ W2 = (const _Float16*)p.w2_frag + lane * 8;   // frag32_w2:    [NCH][NT][2][64][8]
    auto frag_src = [&](int ch, int f) { return f < KS ? W1 + (size_t)(ch * KS + f) * 512 : W2 + (size_t)(ch * 2 * NT + (f - KS)) * 512; };
    auto stage = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int f = wv * NFW + i;
            __builtin_amdgcn_global_load_lds((const void*)frag_src(ch, f), (__attribute__((address_space(3))) void*)(wbuf(ch) + (size_t)f * 1024), 16, 0, 0);
        }
    };
    auto lds_frag = [&](int ch, int j) { return *(const half8*)(wbuf(ch) + (size_t)j * 1024 + lane * 16); };   // consumption order = storage order
#if W2X_MLP2Q_EARLY0
    stage(0);                                  // the oldest requests of the wave: they land under the row fetch
#endif
    for (int i = tid; i < 3 * C; i += K::NWV * 64) ((float*)(smem + BIASQ))[i] = i < 2 * C ? p.b1[i] : p.b2[i - 2 * C];
    const float* B1s = (const float*)(smem + BIASQ) + h * 4;
    const float* B2s = B1s + 2 * C;

    // ---- x rows: flat coalesced pieces -> slab -> LayerNorm in fragment layout (lane (r32, h): channels ks*16 + 8h .. +7 of row r32)
    {
        half8 xr[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) xr[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vl + k * 1024u, 0, 0));
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            *(half8*)(Xw + r * LDX + c * 8) = xr[k];
        }
    }
    W2X_PHASE_FENCE();
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
#if W2X_MLP2Q_PIPE
    // ---- software-pipelined chunk loop (round 4).  Per chunk a wave issues 24 matrix instructions (12 first-layer, 12 second-layer) and ~130 vector
    // instructions (the GELU of 16 values per lane), and as written above they come in separate runs: a 32 x 32 product lets about four vector
    // instructions issue for free while it occupies the pipe (tools/issue_model.hip: 23.4 ticks alone, 31.6 with eight), but a run of products has none to
    // hide and the GELU run has no product to hide behind.  Here the FIRST-layer products of chunk c + 1 (they depend on nothing of chunk c) are issued
    // one by one between the pieces of chunk c's GELU.  What it takes: a second first-layer accumulator (16 registers), and the staged chunks skewed by
    // half a chunk - buffer c holds W1 of chunk c + 1 and W2 of chunk c (fragments 0 .. KS-1 / KS .. NF-1, consumption order = storage order as before).
    W2X_PHASE_FENCE();
    auto skew_src = [&](int c, int f) { return f < KS ? W1 + (size_t)((c + 1) * KS + f) * 512 : W2 + (size_t)(c * 2 * NT + (f - KS)) * 512; };
    auto stage_skew = [&](int c) {               // c = -1: only W1 of chunk 0; c = NCH - 1: only W2 of the last chunk
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int f = wv * NFW + i;
            if ((f < KS && c + 1 >= NCH) || (f >= KS && c < 0)) continue;
            __builtin_amdgcn_global_load_lds((const void*)skew_src(c, f), (__attribute__((address_space(3))) void*)(wbuf(c & 1) + (size_t)f * 1024), 16, 0, 0);
        }
    };
    __syncthreads();                           // every wave holds its rows in registers: the slab area becomes weight buffers
    stage_skew(-1);
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
    __syncthreads();
    stage_skew(0);                             // lands under the first chunk's first-layer products
    float16v acc1n;
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *(const float4v*)(B1s + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1n[4 * q + j] = b[j];
        }
        half8 w0[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i) w0[i] = lds_frag(-1, i);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            acc1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[j % RING], xreg[j], acc1n, 0, 0, 0);
            if (j + RING < KS) { w0[j % RING] = lds_frag(-1, j + RING); W2X_RING_FENCE(); }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's share of buffer 0
    __syncthreads();
    half8 wr[RING];
    float16v acc2[NT];                         // rows = output channels 32nt + 8q + 4h + j in register 4q + j, columns = tokens; from b2
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *(const float4v*)(B2s + nt * 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[nt][4 * q + j] = b[j];
        }
    half8 xres[NP];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const bool more = ch + 1 < NCH;        // there is a next chunk whose first-layer products run under this chunk's GELU
        if (more) stage_skew(ch + 1);
        const int f0 = more ? 0 : KS;          // first fragment consumed from this buffer
#pragma unroll
        for (int i = 0; i < RING; ++i) wr[i] = lds_frag(ch, f0 + i);
        const float16v acc1 = acc1n;
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v b = *(const float4v*)(B1s + (ch + 1) * 32 + q * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1n[4 * q + j] = b[j];
            }
        }
        // GELU of chunk ch in eight pieces of one value pair each, a first-layer product of chunk ch + 1 in front of every piece (and of the
        // four conversions); the scheduling barriers keep the pieces between the products
        float2v gp[8];
        half8 a2[2];
        // ten pieces - pairs 0..3, pack a2[0], pairs 4..7, pack a2[1] - spread over the KS product slots: slot j takes pieces [10 j / KS, 10 (j + 1) / KS)
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (more) {
                acc1n = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[j % RING], xreg[j], acc1n, 0, 0, 0);
                wr[j % RING] = lds_frag(ch, j + RING);          // (f0 = 0 here: fragments KS .. KS + RING - 1 are the first of the second layer)
                W2X_RING_FENCE();
            }
#pragma unroll
            for (int e = 10 * j / KS; e < 10 * (j + 1) / KS; ++e) {
                if (e == 4 || e == 9) {
                    const int s2 = e == 4 ? 0 : 1;
                    a2[s2] = (half8){(_Float16)gp[4 * s2][0], (_Float16)gp[4 * s2][1], (_Float16)gp[4 * s2 + 1][0], (_Float16)gp[4 * s2 + 1][1],
                                     (_Float16)gp[4 * s2 + 2][0], (_Float16)gp[4 * s2 + 2][1], (_Float16)gp[4 * s2 + 3][0], (_Float16)gp[4 * s2 + 3][1]};
                } else {
                    const int w = e < 4 ? e : e - 1;             // value pair w = registers 2w, 2w + 1 of the accumulator
                    gp[w] = gelu_fast2((float2v){acc1[2 * w], acc1[2 * w + 1]});
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch == NCH - 1) {   // the residual rows, requested as soon as the normalised copies have served their last product
#pragma unroll
            for (int k = 0; k < NP; ++k) xres[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vl + k * 1024u, 0, W2X_LD_LAST_AUX));
        }
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {     // second layer: fragment KS + i = (output tile i >> 1, k-step i & 1)
            const int k = (more ? KS : 0) + i; // position in this chunk's consumption order
            acc2[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[k % RING], a2[i & 1], acc2[i >> 1], 0, 0, 0);
            if (f0 + k + RING < NF) { wr[k % RING] = lds_frag(ch, f0 + k + RING); W2X_RING_FENCE(); }
        }
        if (more) __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
#else
    W2X_PHASE_FENCE();
#if W2X_MLP2Q_EARLY0
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's share of chunk 0 (requested first) has landed
    __syncthreads();                           // every wave holds its rows in registers: the slab area becomes the odd chunks' buffer
#else
    __syncthreads();                           // every wave holds its rows in registers: the slab area becomes weight buffers
    stage(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's share of chunk 0 has landed
    __syncthreads();
#endif

    half8 wr[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) wr[i] = lds_frag(0, i);
    float16v acc2[NT];                         // rows = output channels 32nt + 8q + 4h + j in register 4q + j, columns = tokens; from b2
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *(const float4v*)(B2s + nt * 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[nt][4 * q + j] = b[j];
        }
    half8 xres[NP];
#if defined(W2X_MLP2Q_PRIO) && W2X_MLP2Q_PRIO == 1   // s_setprio by phase: 1 = the chunk loop at priority 1, row phases at 0; 2 = the reverse; 3 = only the GELU at 1
    __builtin_amdgcn_s_setprio(1);
#elif defined(W2X_MLP2Q_PRIO) && W2X_MLP2Q_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH && !(W2X_MLP2Q_EXP & 2)) stage(ch + 1);
        float16v acc1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v b = *(const float4v*)(B1s + ch * 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[4 * q + j] = b[j];
        }
#if W2X_MLP2Q_SPLITACC
        float16v acc1b = {};                   // odd k-steps on a second accumulator: two chains of six dependent products instead of one of twelve
#endif
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#if W2X_MLP2Q_SPLITACC
            if (j & 1) acc1b = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[j % RING], xreg[j], acc1b, 0, 0, 0);
            else
#endif
            if (!(W2X_MLP2Q_EXP & 8)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[j % RING], xreg[j], acc1, 0, 0, 0);
            else acc1[j % 16] += (float)wr[j % RING][0] + (float)xreg[j][0];
            wr[j % RING] = lds_frag(ch, j + RING);
            W2X_RING_FENCE();
        }
#if W2X_MLP2Q_SPLITACC
        acc1 += acc1b;
#endif
        if (ch == NCH - 1) {   // the residual rows, requested as soon as the normalised copies have served their last product
#pragma unroll
            for (int k = 0; k < NP; ++k) xres[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vl + k * 1024u, 0, W2X_LD_LAST_AUX));
        }
        half8 a2[2];
#if defined(W2X_MLP2Q_PRIO) && W2X_MLP2Q_PRIO == 3
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            auto act2 = [](float2v v) { return (W2X_MLP2Q_EXP & 4) ? v : gelu_fast2(v); };
            const float2v g0 = act2((float2v){acc1[8 * s2 + 0], acc1[8 * s2 + 1]});
            const float2v g1 = act2((float2v){acc1[8 * s2 + 2], acc1[8 * s2 + 3]});
            const float2v g2 = act2((float2v){acc1[8 * s2 + 4], acc1[8 * s2 + 5]});
            const float2v g3 = act2((float2v){acc1[8 * s2 + 6], acc1[8 * s2 + 7]});
            a2[s2] = (half8){(_Float16)g0[0], (_Float16)g0[1], (_Float16)g1[0], (_Float16)g1[1],
                             (_Float16)g2[0], (_Float16)g2[1], (_Float16)g3[0], (_Float16)g3[1]};
        }
#if defined(W2X_MLP2Q_PRIO) && W2X_MLP2Q_PRIO == 3
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {     // fragment KS + i = (output tile i >> 1, k-step i & 1)
            const int j = KS + i;
            if (!(W2X_MLP2Q_EXP & 8)) acc2[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[j % RING], a2[i & 1], acc2[i >> 1], 0, 0, 0);
            else acc2[i >> 1][i % 16] += (float)wr[j % RING][0] + (float)a2[i & 1][0];
            if (j + RING < NF) { wr[j % RING] = lds_frag(ch, j + RING); W2X_RING_FENCE(); }
        }
        if (ch + 1 < NCH && !(W2X_MLP2Q_EXP & 2)) __builtin_amdgcn_s_waitcnt(0x0F70);
        if (!(W2X_MLP2Q_EXP & 1)) __syncthreads();
        if (ch + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < RING; ++i) wr[i] = lds_frag(ch + 1, i);
        }
    }
#if defined(W2X_MLP2Q_PRIO) && W2X_MLP2Q_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif defined(W2X_MLP2Q_PRIO) && W2X_MLP2Q_PRIO == 2
    __builtin_amdgcn_s_setprio(1);
#endif

#endif
    W2X_PHASE_FENCE();
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(half4*)(Xw + r32 * LDX + nt * 32 + q * 8 + h * 4) = (half4){(_Float16)acc2[nt][4 * q], (_Float16)acc2[nt][4 * q + 1], (_Float16)acc2[nt][4 * q + 2], (_Float16)acc2[nt][4 * q + 3]};
    W2X_PHASE_FENCE();
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
        const half8 o = *(const half8*)(Xw + r * LDX + c * 8) + xres[k];
        if (!(W2X_MLP2Q_EXP & 32) || o[0] == (_Float16)12345.f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), YB, vo + k * 1024u, 0, 0);
        if (p.stats_out) *(half8*)(Xw + r * LDX + c * 8) = o;
    }
    W2X_PHASE_FENCE();
    if (p.stats_out && lane < nrows) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int c = 0; c < PPR; ++c) sum_sq8(*(const half8*)(Xw + lane * LDX + c * 8), s, q);
        const float mean = s * (1.f / C);
        p.stats_out[2 * (row0 + lane)] = mean;
        p.stats_out[2 * (row0 + lane) + 1] = __builtin_amdgcn_rsqf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out);
    }
}

template <int C, int NW>
hipError_t launch_mlp2q_c(const MlpParams& p, hipStream_t s) {
    using K = Mlp2Cfg<C, 2, NW>;
    constexpr int SMEMQ = W2X_MLP2Q_EARLY0 ? K::NWV * K::SLAB + K::WBUF + 3 * C * 4 : K::SMEM;
    static unsigned lds_ok = 0;
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp2q_kernel<C, NW>, SMEMQ, lds_ok); e != hipSuccess) return e;
    const long max_rows = (long)((0xFFF00000u / (C * 2)) / K::BM) * K::BM;
    for (long r0 = 0; r0 < p.M; r0 += max_rows) {
        MlpParams q = p;
        q.M = p.M - r0 < max_rows ? p.M - r0 : max_rows;
        q.x = (const char*)p.x + (size_t)r0 * C * 2; q.y = (char*)p.y + (size_t)r0 * C * 2;
        if (p.stats_out) q.stats_out = p.stats_out + 2 * r0;
        dim3 grid((unsigned)((q.M + K::BM - 1) / K::BM));
        hipLaunchKernelGGL((mlp2q_kernel<C, NW>), grid, dim3(K::NWV * 64), SMEMQ, s, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int C, int TT, int NW>
hipError_t launch_mlp2_c(const MlpParams& p, hipStream_t s) {
    using K = Mlp2Cfg<C, TT, NW>;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp2_kernel<C, TT, NW>, K::SMEM, lds_ok); e != hipSuccess) return e;
    // the kernel addresses x / y with 32-bit byte offsets: longer passes run in pieces of whole workgroups
    const long max_rows = (long)((0xFFF00000u / (C * 2)) / K::BM) * K::BM;
    for (long r0 = 0; r0 < p.M; r0 += max_rows) {
        MlpParams q = p;
        q.M = p.M - r0 < max_rows ? p.M - r0 : max_rows;
        q.x = (const char*)p.x + (size_t)r0 * C * 2; q.y = (char*)p.y + (size_t)r0 * C * 2;
        if (p.stats_out) q.stats_out = p.stats_out + 2 * r0;
        dim3 grid((unsigned)((q.M + K::BM - 1) / K::BM));
        hipLaunchKernelGGL((mlp2_kernel<C, TT, NW>), grid, dim3(K::NWV * 64), K::SMEM, s, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

bool mlp96q_supported(const MlpParams& p);                        // k_mlp96q.hip: C = 96 with both matrices resident in LDS
hipError_t launch_mlp96q(const MlpParams& p, hipStream_t s);

hipError_t launch_mlp2(const MlpParams& p, hipStream_t s) {
    // C = 96: the resident-weight kernel on 32x32 tiles (k_mlp96q.hip; the engine stores its weights in that kernel's fragment order,
    // fragorder.h frag32_*).  This file's chunked schedule at C = 96 (frag_major / frag_w2 order) remains for tools/ab/mlp_ab.hip.
    // C = 192 (weights 288 KiB): shared-weight schedule, 32 rows per wave, 4 waves per workgroup.
    // Measured alternatives: 8 waves per workgroup (a chunk staged once per 256 rows, one workgroup per CU) 2.0 ms of C = 192 MLP time
    // per frame against 1.77; in round 1 6 / 12 waves per workgroup 2.45 / 1.86 ms of MLP time per frame against 1.58; a per-wave
    // register ring straight from L2 (TT = 4) and 64 rows per wave with shared weights were slower as well.
    if (p.C == 96 && p.frag32) return mlp96q_supported(p) ? launch_mlp96q(p, s) : hipErrorInvalidValue;
    if (p.C == 96) return launch_mlp2_c<96, 2, 4>(p, s);
    if (p.C == 192 && p.frag32) return launch_mlp2q_c<192, 4>(p, s);       // weights in the 32x32x16 fragment order (the engine)
    if (p.C == 192) return launch_mlp2_c<192, W2X_MLP192_TT, W2X_MLP192_NW>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace w2x

