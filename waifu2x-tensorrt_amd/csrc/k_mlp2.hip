// Fused transformer MLP branch for gfx950, wave-private rows:   y = x + W2 * gelu(W1 * LayerNorm(x) + b1) + b2
// Every wave owns 16*TT token rows from the first load to the last store:
//   * x rows arrive as one flat, fully coalesced stream into the wave's LDS slab; each lane then reads the 16-byte pieces it
//     uses as MFMA fragments, LayerNorm statistics are the lane's own sums plus two row swaps, and the normalised pieces are
//     the operand registers of GEMM1 as they stand (LDS program order is the only ordering needed inside a wave, see
//     W2X_PHASE_FENCE);
//   * weights (SHARE, the shipped schedule): the four waves of a workgroup stage each 32-hidden-unit chunk once into LDS,
//     fragment-major (engine.cpp: one contiguous KiB per wave load), double-buffered, one workgroup barrier per chunk - the
//     per-wave weight stream from L2 was 56 % of the C = 192 kernel's time.  (!SHARE: a per-wave register ring straight from L2,
//     kept for A/B runs);
//   * GEMM1 is computed transposed (rows = 32 hidden units of the chunk, columns = tokens) so that the GELU'd accumulators
//     are, as they stand, the B fragments of GEMM2 with a permuted k order (W2 is stored with the same permutation) - no LDS
//     round trip between the two products; GEMM2 is transposed too (rows = output channels), so a lane ends with 4 consecutive
//     channels of a token: b2 is the initial accumulator and the tile reaches the slab in 8-byte stores;
//   * the result tile goes through the slab once so that residual add and stores are flat 16-byte pieces again; at C = 96 the
//     raw x rows are still in the slab at that point (the weight buffers sit behind the slabs), so x is read from HBM once.
// TT = 2 for both widths, which leaves C = 96 at 3 waves per SIMD.
#include "kernels.h"

#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// GELU(x) = max(x,0) - 0.5 u 2^-q(u), u = min(|x|, 6.5): see k_mlp.hip / tools/fit_gelu.py (|err| < 3.2e-7)
// Two values at a time: the polynomial, the products and the final fma are v_pk_*_f32 (one issue slot for both values);
// min / max / exp2 have no packed form.  Same operations per element as the scalar form, so the results are identical.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v splat2(float c) { return (float2v){c, c}; }
__device__ __forceinline__ float2v gelu_fast2(float2v x) {
    const float2v u = {fminf(fabsf(x[0]), 6.5f), fminf(fabsf(x[1]), 6.5f)};
    float2v q = __builtin_elementwise_fma(splat2(-2.992485764e-05f), u, splat2(7.398797018e-04f));
    q = __builtin_elementwise_fma(q, u, splat2(-7.977479093e-03f));
    q = __builtin_elementwise_fma(q, u, splat2(5.323820859e-02f));
    q = __builtin_elementwise_fma(q, u, splat2(4.589156733e-01f));
    q = __builtin_elementwise_fma(q, u, splat2(1.151147085e+00f));
    const float2v t = q * u;
    const float2v e = {__builtin_amdgcn_exp2f(-t[0]), __builtin_amdgcn_exp2f(-t[1])};
    const float2v m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(splat2(-0.5f) * u, e, m);
}
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
// values on the second wave of a SIMD at full problem size (tools/mlp_ab.hip reproduces it: first workgroup per CU always
// right, co-resident ones wrong, not cured by s_waitcnt / s_barrier) and was replaced by register swaps.
#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")

template <int C, int TT, int NW, bool SHARE>
struct Mlp2Cfg {
    static constexpr int RW = 16 * TT;           // rows per wave
    static constexpr int NWV = NW;               // independent waves per workgroup
    static constexpr int BM = NWV * RW;          // rows per workgroup
    static constexpr int LDX = C + 8;            // slab row stride in halves: 16-byte pieces rotate over the banks
    static constexpr int PPR = C / 8;            // 16-byte pieces per row
    static constexpr int KS = C / 32;            // k-steps of GEMM1
    static constexpr int NT = C / 16;            // output n-tiles of GEMM2
    static constexpr int NCH = 2 * C / 32;       // hidden chunks of 32
    static constexpr int NP = RW * PPR / 64;     // flat 16-byte pieces per lane
    static constexpr int SLAB = RW * LDX * 2;    // bytes per wave
    static constexpr int NF = 2 * KS + NT;       // weight fragments (KiB) per hidden chunk
    static constexpr int WBUF = NF * 1024;       // SHARE: one staged chunk, two buffers
    // KEEP (C = 96): the weight buffers sit behind the slabs (50.6 KB per workgroup, still 3 workgroups per CU), so the raw x rows stay
    // in the slab and serve the residual add - x is read from HBM once.  Otherwise the buffers alias the slabs (x lives in
    // registers by then) and the residual rows are fetched a second time.
    static constexpr bool KEEP = SHARE && C == 96;
    static constexpr int SMEM = KEEP ? NWV * SLAB + 2 * WBUF : SHARE ? (NWV * SLAB > 2 * WBUF ? NWV * SLAB : 2 * WBUF) : NWV * SLAB;
    static_assert(!SHARE || NF % NWV == 0, "fragments per wave");
    static_assert(RW * PPR % 64 == 0, "flat piece count");
};

template <int C, int TT, int NW, bool SHARE>
// (the second launch bound is hipcc's minimum number of waves per SIMD, not blocks per CU)
__global__ __launch_bounds__(NW * 64, (SHARE && TT <= 2 && C == 96 ? 3 : (NW >= 8 ? 1 : 8 / NW))) void mlp2_kernel(const MlpParams p) {
    using K = Mlp2Cfg<C, TT, NW, SHARE>;
    constexpr int RW = K::RW, LDX = K::LDX, PPR = K::PPR, KS = K::KS, NT = K::NT, NCH = K::NCH, NP = K::NP;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* Xw = (_Float16*)(smem + wv * K::SLAB);          // [RW][LDX]

    const long row0 = ((long)blockIdx.x * K::NWV + wv) * RW;       // first row of this wave
    const long nrows = p.M - row0 < RW ? p.M - row0 : RW;     // may be <= 0: the wave then only runs dead arithmetic
    const int npieces = nrows > 0 ? (int)nrows * PPR : 0;
    const _Float16* __restrict__ X = (const _Float16*)p.x + row0 * C;
    const _Float16* __restrict__ W1 = (const _Float16*)p.w1_frag + lane * 8;   // [NCH*2 row tiles][KS][64][8]
    const _Float16* __restrict__ W2 = (const _Float16*)p.w2_frag + lane * 8;   // [NCH][NT][64][8], k order of the GELU'd accumulators

    // ---- weights.  !SHARE: per-wave register ring, chunk 0 now.  SHARE: the four waves stage each chunk once into LDS
    //      (two buffers over the slab area; wave w brings fragments w*NFW .. of the next chunk while the current one is consumed)
    constexpr int NF = K::NF, NFW = NF / K::NWV;
    half8 w1r[2 * KS], w2r[NT];
    half8 stg[NFW];
    _Float16* WB = (_Float16*)(smem + (K::KEEP ? K::NWV * K::SLAB : 0));
    auto frag_src = [&](int ch, int f) { return f < 2 * KS ? W1 + (size_t)(ch * 2 * KS + f) * 512 : W2 + (size_t)(ch * NT + (f - 2 * KS)) * 512; };
    if (SHARE) {
#pragma unroll
        for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(0, wv * NFW + i);
    } else {
#pragma unroll
        for (int f = 0; f < 2 * KS; ++f) w1r[f] = *(const half8*)(W1 + (size_t)f * 512);
#pragma unroll
        for (int f = 0; f < NT; ++f) w2r[f] = *(const half8*)(W2 + (size_t)f * 512);
    }

    // ---- x rows: flat coalesced load -> slab
    {
        half8 xr[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane;
            half8 h = {};
            if (idx < npieces) h = *(const half8*)(X + (size_t)idx * 8);
            xr[k] = h;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
            *(half8*)(Xw + r * LDX + c * 8) = xr[k];
        }
    }
    W2X_PHASE_FENCE();
    // ---- LayerNorm in fragment layout: lane (fr, g) holds channels ks*32 + 8g .. +7 of row 16tt + fr, so the row sums are the
    //      lane's own KS pieces plus the three other lane groups (two row swaps).  SHARE: the normalised pieces are the operand
    //      registers of GEMM1 as they stand; otherwise they go back in place (rows without data hold zeros: 0 * rstd - 0)
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
        const float rstd = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps);
        const float nm = -mean * rstd;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (SHARE) xreg[tt][ks] = norm8(raw[ks], rstd, nm);
            else *(half8*)(Xw + (tt * 16 + fr) * LDX + ks * 32 + g * 8) = norm8(raw[ks], rstd, nm);
        }
    }
    W2X_PHASE_FENCE();
    if (SHARE) {
        if (!K::KEEP) __syncthreads();         // every wave holds its rows in registers: the slab area becomes weight buffers
#pragma unroll
        for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
        __syncthreads();
    }
    // GEMM2 is computed transposed as well (rows = output channels, columns = tokens): a lane ends with 4 consecutive channels of
    // one token (8-byte slab stores) and b2 is the initial accumulator
    float4v acc2[TT][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float4v b2v = *(const float4v*)(p.b2 + j * 16 + g * 4);
#pragma unroll
        for (int i = 0; i < TT; ++i) acc2[i][j] = b2v;
    }

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {   // fully unrolled: the ring registers are renamed statically
        // GEMM1 (transposed): acc1[ht][tt] = W1[32ch + 16ht ..][:] * Xn[16tt ..][:]^T   (rows = hidden, columns = tokens)
        const _Float16* wcur = WB + (size_t)(ch & 1) * (K::WBUF / 2) + lane * 8;      // SHARE: this chunk's fragments in LDS
        if (SHARE && ch + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(ch + 1, wv * NFW + i);
        }
        float4v acc1[2][TT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc1[i][tt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 xb[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) xb[tt] = SHARE ? xreg[tt][ks] : *(const half8*)(Xw + (tt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                const half8 wa = SHARE ? *(const half8*)(wcur + (size_t)(ht * KS + ks) * 512) : w1r[ht * KS + ks];
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc1[ht][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[tt], acc1[ht][tt], 0, 0, 0);
                if (!SHARE && ch + 1 < NCH) w1r[ht * KS + ks] = *(const half8*)(W1 + (size_t)(((ch + 1) * 2 + ht) * KS + ks) * 512);
            }
        }
        // bias + GELU in place; lane holds hidden rows 16ht + 4g + j of token column fr -> B fragment of GEMM2 for the
        // k order (ht 0: slots 0..3, ht 1: slots 4..7)
        half8 a2[TT];
        {
            const float4v be = *(const float4v*)(p.b1 + ch * 32 + g * 4);
            const float4v bo = *(const float4v*)(p.b1 + ch * 32 + 16 + g * 4);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const float4v e = acc1[0][tt], o = acc1[1][tt];
                const float2v g0 = gelu_fast2((float2v){e[0], e[1]} + (float2v){be[0], be[1]});
                const float2v g1 = gelu_fast2((float2v){e[2], e[3]} + (float2v){be[2], be[3]});
                const float2v g2 = gelu_fast2((float2v){o[0], o[1]} + (float2v){bo[0], bo[1]});
                const float2v g3 = gelu_fast2((float2v){o[2], o[3]} + (float2v){bo[2], bo[3]});
                a2[tt] = (half8){(_Float16)g0[0], (_Float16)g0[1], (_Float16)g1[0], (_Float16)g1[1],
                                 (_Float16)g2[0], (_Float16)g2[1], (_Float16)g3[0], (_Float16)g3[1]};
            }
        }
        // GEMM2 (transposed): acc2[tt][nt] += W2[16nt ..][chunk] * H[tokens][chunk]^T
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const half8 wb = SHARE ? *(const half8*)(wcur + (size_t)(2 * KS + nt) * 512) : w2r[nt];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc2[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, a2[tt], acc2[tt][nt], 0, 0, 0);
            if (!SHARE && ch + 1 < NCH) w2r[nt] = *(const half8*)(W2 + (size_t)((ch + 1) * NT + nt) * 512);
        }
        if (SHARE) {   // hand the next chunk over: the other buffer was last read one iteration ago, before the previous barrier
            if (ch + 1 < NCH) {
#pragma unroll
                for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)((ch + 1) & 1) * (K::WBUF / 2) + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
            }
            __syncthreads();
        }
    }

    W2X_PHASE_FENCE();
    // ---- epilogue: residual pieces first (KEEP: from the slab, which still holds the raw rows; otherwise a second fetch),
    //      accumulators -> fp16 tile in the slab, then flat pieces
    half8 xres[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
        half8 h = {};
        if (K::KEEP) h = *(const half8*)(Xw + r * LDX + c * 8);
        else if (idx < npieces) h = *(const half8*)(X + (size_t)idx * 8);
        xres[k] = h;
    }
    W2X_PHASE_FENCE();
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
            *(half4*)(Xw + (tt * 16 + fr) * LDX + nt * 16 + g * 4) = (half4){(_Float16)acc2[tt][nt][0], (_Float16)acc2[tt][nt][1], (_Float16)acc2[tt][nt][2], (_Float16)acc2[tt][nt][3]};
    W2X_PHASE_FENCE();
    _Float16* __restrict__ Y = (_Float16*)p.y + row0 * C;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = k * 64 + lane, r = idx / PPR, c = idx - r * PPR;
        const half8 o = *(const half8*)(Xw + r * LDX + c * 8) + xres[k];     // fp16 + fp16 rounded once == fp32 add rounded to fp16
        if (idx < npieces) *(half8*)(Y + (size_t)idx * 8) = o;
        if (p.stats_out) *(half8*)(Xw + r * LDX + c * 8) = o;
    }
    W2X_PHASE_FENCE();
    if (p.stats_out && lane < nrows) {   // LayerNorm statistics of the produced rows for an un-fused consumer
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int c = 0; c < PPR; ++c) sum_sq8(*(const half8*)(Xw + lane * LDX + c * 8), s, q);
        const float mean = s * (1.f / C);
        p.stats_out[2 * (row0 + lane)] = mean;
        p.stats_out[2 * (row0 + lane) + 1] = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out);
    }
}

template <int C, int TT, int NW, bool SHARE>
hipError_t launch_mlp2_c(const MlpParams& p, hipStream_t s) {
    using K = Mlp2Cfg<C, TT, NW, SHARE>;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)mlp2_kernel<C, TT, NW, SHARE>, K::SMEM, lds_ok); e != hipSuccess) return e;
    dim3 grid((unsigned)((p.M + K::BM - 1) / K::BM));
    hipLaunchKernelGGL((mlp2_kernel<C, TT, NW, SHARE>), grid, dim3(K::NWV * 64), K::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

bool mlp96p_supported(const MlpParams& p);                        // k_mlp96p.hip: C = 96 with both matrices resident in LDS
hipError_t launch_mlp96p(const MlpParams& p, hipStream_t s);

hipError_t launch_mlp2(const MlpParams& p, hipStream_t s) {
    // C = 96: the resident-weight kernel (k_mlp96p.hip); W2X_MLP96_CHUNKED=1 (read once per process) keeps this file's chunked
    // schedule for A/B runs.  C = 192 (weights 288 KiB): shared-weight schedule, 32 rows per wave, 4 waves per workgroup.
    // Measured alternatives (git history): 6 / 12 waves per workgroup 2.45 / 1.86 ms of MLP time per frame against 1.58; the
    // per-wave register ring (SHARE = false, TT = 4) and 64 rows per wave with shared weights were slower as well.
    static const bool chunked96 = getenv("W2X_MLP96_CHUNKED") != nullptr;
    if (p.C == 96 && !chunked96 && mlp96p_supported(p)) return launch_mlp96p(p, s);
    if (p.C == 96) return launch_mlp2_c<96, 2, 4, true>(p, s);
    if (p.C == 192) return launch_mlp2_c<192, 2, 4, true>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace w2x
