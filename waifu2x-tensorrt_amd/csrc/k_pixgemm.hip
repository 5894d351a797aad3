// Streaming GEMM kernels for the memory-bound projections of the Swin U-Net (pixel-shuffle up-projection, image head, patch
// merge); gemm_kernel (k_gemm.hip) serves every other shape and remains the reference for these.
//
// Pixel-shuffle projection for gfx950:   out[b][y*r+dy][x*r+dx][:] = x[b][y][x][:] * W[(dy*r+dx)*Cso .. +Cso][:]^T + bias (+ residual)
// i.e. the Swin "patch expand" (Linear C -> r*r*Cso, DepthToSpace, skip add).  Same result as gemm_kernel with omode 2
// (k_gemm.hip), which serves every other shape; this variant exists because these launches are memory-bound (1.25 GB
// moved for 0.1 TFLOP on config 3's last up-projection) and the general kernel's load -> compute -> store phases per
// workgroup keep too few bytes in flight (1.8 TB/s).  Schedule, as in k_mlp2.hip:
//   * a wave owns 32 token rows from load to store; rows arrive as a flat coalesced stream, pass through LDS once and stay
//     in registers as MFMA A fragments for the whole kernel;
//   * the weight matrix streams through LDS in stages of G n-tiles (fragment-major copy, fragorder.h), staged once per
//     workgroup by its four waves, double-buffered, one barrier per stage;
//   * when the Cso columns of one sub-pixel are complete the wave turns them around through its own LDS tile and issues
//     the residual loads and the 16-byte stores for that sub-pixel while the next one is being computed.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")

// How many stages ahead the weight fragments of pixgemm_kernel are requested (each stage in flight holds NFW x 4 registers per lane).  A
// stage is 24 products per wave - 0.2 us - against an L2 round trip of 1 us and more under load, and a workgroup walks 12 to 24 stages
// with a barrier each: one stage ahead (round 2) left most of every round trip exposed.
// 0 = per shape: three ahead for 192 -> 4 x 96 (210 registers), two for the others (192 -> 4 x 192 would spill, cunet's shapes would lose a wave per SIMD).
// m-tiles per wave of the 192 -> 4 x 192 projection: with two, a wave's skip rows (48 registers) cannot travel with its row fragments (48) and
// are fetched when a sub-pixel is complete - two exposed round trips to memory per sub-pixel at two waves per SIMD; with one (16 rows per wave,
// 64 per workgroup, three workgroups per CU) they are requested a sub-pixel ahead like everywhere else.
#ifndef W2X_PIX192_TT
#define W2X_PIX192_TT 1
#endif
#ifndef W2X_PIX_AHEAD
#define W2X_PIX_AHEAD 0
#endif

template <int K, int CSO, int G, int TT_ = 2>
struct PixCfg {
    static constexpr int TT = TT_, RW = 16 * TT, BM = 4 * RW;  // m-tiles / rows per wave, rows per workgroup
    static constexpr int KS = K / 32, NTS = CSO / 16;          // k-steps, n-tiles per sub-pixel
    static constexpr int LDO = CSO + 8, LDXI = K + 8;          // LDS row strides (halves): output tile, input staging
    static constexpr int PPI = K / 8, PPO = CSO / 8;           // 16-byte pieces per input row / output pixel
    static constexpr int NPI = RW * PPI / 64, NPO = RW * PPO / 64;
    static constexpr int NF = G * KS, NFW = NF / 4;            // fragments per stage / per wave
    static constexpr bool GATES = K != 192;                    // squeeze-excite gates on the operands exist in cunet's shapes only (pixgemm_supported); without
                                                               // their branches the stage loop's waits for memory come out exact
    static constexpr int AHEAD = W2X_PIX_AHEAD ? W2X_PIX_AHEAD : (K == 192 && CSO == 96 ? 3 : 2);
    static constexpr int WBUF = NF * 1024;
    static constexpr int OTILE = RW * LDO * 2 + RW * 8;        // per wave: output tile + (out, res) base offsets per row
    static constexpr int IN_BYTES = 4 * RW * LDXI * 2;         // input staging aliases everything (dead before the first stage)
    static constexpr int MAIN = 2 * WBUF + 4 * OTILE;
    static constexpr int BIAS_OFF = MAIN > IN_BYTES ? MAIN : IN_BYTES;      // bias [<= 16 sub-pixels][CSO] fp32 behind everything
    static constexpr int SMEM_MAX = BIAS_OFF + 16 * CSO * 4;   // a launch asks for BIAS_OFF + its own sub-pixels x CSO x 4
    // skip rows: requested all at once when a sub-pixel starts (up to 8 pieces = 32 registers), else in groups when it is complete
    static constexpr bool EARLY = NPO <= 8;
    static constexpr int RGRP = EARLY ? NPO : (NPO % 6 == 0 ? 6 : NPO % 4 == 0 ? 4 : NPO);
    static_assert(NTS % G == 0 && NF % 4 == 0 && RW * PPI % 64 == 0 && RW * PPO % 64 == 0 && NPO % RGRP == 0, "tiling");
};

template <int K, int CSO, int G, int TT_>
__global__ __launch_bounds__(256, TT_ == 1 ? 3 : 2) void pixgemm_kernel(const GemmParams p) {
    using C = PixCfg<K, CSO, G, TT_>;
    constexpr int TT = C::TT, RW = C::RW, KS = C::KS, NTS = C::NTS, LDO = C::LDO, LDXI = C::LDXI, NFW = C::NFW;
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* WB = (_Float16*)smem;                                          // [2][NF][64][8]
    _Float16* Ot = (_Float16*)(smem + 2 * C::WBUF + wv * C::OTILE);           // [RW][LDO]
    unsigned* Rb = (unsigned*)(smem + 2 * C::WBUF + wv * C::OTILE + RW * LDO * 2);   // [RW][2] byte offsets of sub-pixel (0,0): out, res (0xFFFFFFFF: no row)
    float* Bs = (float*)(smem + C::BIAS_OFF);                                 // bias [r * r * CSO]: read per n-tile through LDS
    _Float16* Xin = (_Float16*)(smem + wv * RW * LDXI * 2);                   // input staging (aliases the weight buffers and tiles)

    const long M = (long)p.B * p.Mrows;
    const long row0 = ((long)blockIdx.x * 4 + wv) * RW;
    const long nrows = M - row0 < RW ? M - row0 : RW;
    const _Float16* __restrict__ Wf = (const _Float16*)p.wt_frag + lane * 8;  // [N/16][KS][64][8]
    const int r = p.r, nsub = r * r, nstage = nsub * NTS / G;
    // Rows, residual rows and output rows go through buffer resources (32-bit byte offsets, bounds-checked: an offset of 0xFFFFFFFF -
    // a row that does not exist - reads zeros and drops the store), so no piece needs a predicate; launch_pix() refuses maps of 4 GB
    // and more (the general kernel takes them).
    const __amdgpu_buffer_rsrc_t XB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a.p), 0, (unsigned)(M * (K * 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t OB = __builtin_amdgcn_make_buffer_rsrc(p.out.p, 0, (unsigned)((size_t)p.B * p.out.Hs * p.out.Ws * CSO * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t RB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res.p ? p.res.p : p.out.p), 0,
                                                                         p.res.p ? (unsigned)((size_t)p.B * p.res.Hs * p.res.Ws * p.res.Cs * 2) : 0u, 0x00020000);

    // ---- stage 0 of the weights is requested first, then the rows
    constexpr int D = C::AHEAD;                    // stages in flight; stage st travels in register set st % D
    half8 stg[D][NFW];
#pragma unroll
    for (int i = 0; i < NFW; ++i) stg[0][i] = *(const half8*)(Wf + (size_t)(wv * NFW + i) * 512);
    {
        half8 xr[C::NPI];
        const unsigned vo = nrows > 0 ? (unsigned)(row0 * (K * 2)) + lane * 16u : 0xFFFF8000u;
#pragma unroll
        for (int k = 0; k < C::NPI; ++k) xr[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, vo + k * 1024u, 0, 0));
        if (C::GATES && p.a_scale) {   // squeeze-excite gate of the input map: fp16(x * s), the rounding of the in-place pass
#pragma unroll
            for (int k = 0; k < C::NPI; ++k) {
                const int idx = k * 64 + lane, rr = idx / C::PPI, c = idx - rr * C::PPI;
                const float* sc = p.a_scale + (size_t)((row0 + rr) / p.Mrows) * K + c * 8;
                if (rr < nrows) xr[k] = gate::gate8(xr[k], sc);
            }
        }
#pragma unroll
        for (int k = 0; k < C::NPI; ++k) {
            const int idx = k * 64 + lane, rr = idx / C::PPI, c = idx - rr * C::PPI;
            *(half8*)(Xin + rr * LDXI + c * 8) = xr[k];
        }
    }
    W2X_PHASE_FENCE();
    half8 xa[TT][KS];                              // A fragments: lane (row fr of tile tt, g) holds channels ks*32 + 8g .. +7
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xa[tt][ks] = *(const half8*)(Xin + (tt * 16 + fr) * LDXI + ks * 32 + g * 8);
    // per-row output / residual byte offsets of sub-pixel (0,0), computed once
    unsigned my_ob = 0xFFFFFFFFu, my_rb = 0xFFFFFFFFu;
    if (lane < RW) {
        const long gr = row0 + lane;
        if (gr < M) {
            const int b = (int)(gr / p.Mrows), ml = (int)(gr - (long)b * p.Mrows);
            const int oy = ml / p.aW, ox = ml - oy * p.aW;
            my_ob = (unsigned)(((size_t)(b * p.out.Hs + oy * r) * p.out.Ws + ox * r) * CSO * 2);
            if (p.res.p) my_rb = (unsigned)(((size_t)(b * p.res.Hs + oy * r + p.res.y0) * p.res.Ws + ox * r + p.res.x0) * p.res.Cs * 2);
        }
    }
    __syncthreads();                               // all rows are in registers: the staging area becomes weight buffers + tiles
    if (lane < RW) { Rb[2 * lane] = my_ob; Rb[2 * lane + 1] = my_rb; }
    for (int i = tid; i < nsub * CSO; i += 256) Bs[i] = p.bias[i];
#pragma unroll
    for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[0][i];
    // (requests past the last stage repeat it instead of being skipped: with a fixed number of loads in flight on every path the compiler's
    //  wait for the oldest set leaves the younger ones outstanding - behind a branch it waited for all of them)
#pragma unroll
    for (int d = 1; d < D; ++d)
#pragma unroll
        for (int i = 0; i < NFW; ++i) stg[d][i] = *(const half8*)(Wf + (size_t)(min(d, nstage - 1) * C::NF + wv * NFW + i) * 512);
    __syncthreads();

    // The skip rows of a sub-pixel are REQUESTED when its first n-tiles start and consumed when its last ones are done (round 2 loaded
    // each piece where it was added and waited for it: nsub x NPO exposed HBM round trips per wave); the bias comes from LDS (as a
    // global load its in-order wait also waited for the next stage's weight fragments requested just before).
    half8 rres[C::RGRP];
    // One trip of the outer loop = one sub-pixel; its SPS stages are unrolled, so the register set of a stage (SPS is a multiple of D), the
    // stage that requests the skip rows and the one that stores are compile-time facts and every path issues the same loads in the same
    // order - which is what lets the waits for memory be exact counts.
    constexpr int SPS = NTS / G;
    static_assert(SPS % D == 0, "a sub-pixel's stages must be whole rounds of the register sets");
    for (int sg = 0; sg < nsub; ++sg) {
#pragma unroll
      for (int u = 0; u < SPS; ++u) {
        const int st = sg * SPS + u;
        const _Float16* wcur = WB + (size_t)(st & 1) * (C::WBUF / 2) + lane * 8;
        {                                           // set u carried stage st, which reached LDS at the end of stage st - 1
#pragma unroll
            for (int i = 0; i < NFW; ++i) stg[u % D][i] = *(const half8*)(Wf + (size_t)(min(st + D, nstage - 1) * C::NF + wv * NFW + i) * 512);
            W2X_PHASE_FENCE();     // keeps the requests at the top of the stage (the scheduler would sink them to the LDS stores at its end)
        }
        const int nt0 = u * G;                      // first n-tile of the stage inside its sub-pixel
        const int dy = sg / r, dx = sg - dy * r;
        // (more than eight pieces per lane - 192 output columns with two m-tiles per wave: twelve pieces = 48 registers next to 48 of row fragments -
        //  do not fit; such a shape requests them in groups when the sub-pixel is complete.  The shipped 192-column shape runs one m-tile per wave.)
        constexpr bool EARLY = C::EARLY;
        if (EARLY && u == 0) {                      // (no skip connection: the resource is empty and the loads return zeros without touching memory)
            const unsigned rshift = (unsigned)((dy * p.res.Ws + dx) * p.res.Cs * 2);
#pragma unroll
            for (int k = 0; k < C::NPO; ++k) {
                const int idx = k * 64 + lane, rr = idx / C::PPO, c = idx - rr * C::PPO;
                rres[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(RB, __builtin_elementwise_add_sat(Rb[2 * rr + 1], rshift + c * 16u), 0, 0));
            }
        }
#pragma unroll
        for (int t = 0; t < G; ++t) {
            float4v acc[TT];
            const float b = Bs[(sg * NTS + nt0 + t) * 16 + fr];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[tt] = (float4v){b, b, b, b};         // bias = initial accumulator
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 wb = *(const half8*)(wcur + (size_t)(t * KS + ks) * 512);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[tt][ks], wb, acc[tt], 0, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[tt][j];
                    if (p.act == 1) v = v > 0.f ? v : v * p.alpha;
                    Ot[(tt * 16 + g * 4 + j) * LDO + (nt0 + t) * 16 + fr] = (_Float16)v;
                }
        }
        if (u == SPS - 1) {                        // sub-pixel sg complete: residual add and store as 16-byte pieces
            W2X_PHASE_FENCE();
            const unsigned oshift = (unsigned)((dy * p.out.Ws + dx) * CSO * 2), rshift = (unsigned)((dy * p.res.Ws + dx) * p.res.Cs * 2);
            constexpr int GRP = C::RGRP;                  // late variant: a group of pieces requested together, then added and stored
#pragma unroll
            for (int k0 = 0; k0 < C::NPO; k0 += GRP) {
                if (!EARLY) {
#pragma unroll
                    for (int k = k0; k < k0 + GRP; ++k) {
                        const int idx = k * 64 + lane, rr = idx / C::PPO, c = idx - rr * C::PPO;
                        rres[k - k0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(RB, __builtin_elementwise_add_sat(Rb[2 * rr + 1], rshift + c * 16u), 0, 0));
                    }
                }
#pragma unroll
                for (int k = k0; k < k0 + GRP; ++k) {
                    const int idx = k * 64 + lane, rr = idx / C::PPO, c = idx - rr * C::PPO;
                    half8 o = *(const half8*)(Ot + rr * LDO + c * 8);
                    if (p.res.p) {
                        half8 rv = rres[EARLY ? k : k - k0];
                        if (C::GATES && p.res_scale && rr < nrows) rv = gate::gate8(rv, p.res_scale + (size_t)((row0 + rr) / p.Mrows) * p.res.Cs + c * 8);   // gated skip connection
                        o += rv;                                   // fp16 + fp16 rounded once == fp32 add rounded to fp16
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, o), OB, __builtin_elementwise_add_sat(Rb[2 * rr], oshift + c * 16u), 0, W2X_ST_AUX);
                }
                W2X_PHASE_FENCE();
            }
        }
        {                                           // (after the last stage: a copy nobody reads)
#pragma unroll
            for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)((st + 1) & 1) * (C::WBUF / 2) + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[(u + 1) % D][i];
        }
        __syncthreads();
      }
    }
}

template <int K, int CSO, int G, int TT = 2>
hipError_t launch_pix(const GemmParams& p, hipStream_t s) {
    using C = PixCfg<K, CSO, G, TT>;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)pixgemm_kernel<K, CSO, G, TT>, C::SMEM_MAX, lds_ok); e != hipSuccess) return e;
    const long M = (long)p.B * p.Mrows;
    dim3 grid((unsigned)((M + C::BM - 1) / C::BM));
    hipLaunchKernelGGL((pixgemm_kernel<K, CSO, G, TT>), grid, dim3(256), C::BIAS_OFF + p.r * p.r * CSO * 4, s, p);
    return hipGetLastError();
}

// Image head: Linear 96 -> 64 = 4x4 sub-pixels x 4 stored channels, DepthToSpace(4), Clip.  The whole weight matrix is 12
// fragments, so every wave keeps it in registers (no LDS staging, no barrier) and owns 64 rows; an n-tile holds the four
// horizontally adjacent output pixels of one output row, so a 16-byte piece of the tile is two finished pixels.
__global__ __launch_bounds__(256, 2) void toimage_kernel(const GemmParams p) {
    constexpr int K = 96, KS = 3, TT = 4, RW = 64, N = 64, NT = 4, LDXI = K + 8, LDO = N + 8, PPI = K / 8, NPI = RW * PPI / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* Xin = (_Float16*)(smem + wv * RW * LDXI * 2);   // [RW][LDXI] rows; later the output tile [RW][LDO]
    const long M = (long)p.B * p.Mrows;
    const long row0 = ((long)blockIdx.x * 4 + wv) * RW;
    const long nrows = M - row0 < RW ? M - row0 : RW;
    const int npieces = nrows > 0 ? (int)nrows * PPI : 0;
    const _Float16* __restrict__ X = (const _Float16*)p.a.p + row0 * K;
    const _Float16* __restrict__ Wf = (const _Float16*)p.wt_frag + lane * 8;
    half8 wf[NT][KS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[t][ks] = *(const half8*)(Wf + (size_t)(t * KS + ks) * 512);
    {
        half8 xr[NPI];
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            const int idx = k * 64 + lane;
            half8 h = {};
            if (idx < npieces) h = *(const half8*)(X + (size_t)idx * 8);
            xr[k] = h;
        }
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            const int idx = k * 64 + lane, rr = idx / PPI, c = idx - rr * PPI;
            *(half8*)(Xin + rr * LDXI + c * 8) = xr[k];
        }
    }
    W2X_PHASE_FENCE();
    float4v acc[TT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float b = p.bias[t * 16 + fr];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt][t] = (float4v){b, b, b, b};
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 xa = *(const half8*)(Xin + (tt * 16 + fr) * LDXI + ks * 32 + g * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[tt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, wf[t][ks], acc[tt][t], 0, 0, 0);
        }
    W2X_PHASE_FENCE();
    _Float16* Ot = Xin;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[tt][t][j];
                if (p.has_clip) v = fminf(fmaxf((float)(_Float16)v, p.clip_lo), p.clip_hi);
                Ot[(tt * 16 + g * 4 + j) * LDO + t * 16 + fr] = (_Float16)v;
            }
    W2X_PHASE_FENCE();
    _Float16* __restrict__ Og = (_Float16*)p.out.p;
    // 8 pieces per row: piece c = output row dy = c / 2, pixels dx = 2 * (c & 1), +1
#pragma unroll
    for (int k = 0; k < RW * 8 / 64; ++k) {
        const int idx = k * 64 + lane, rr = idx >> 3, c = idx & 7;
        const long gr = row0 + rr;
        if (gr < M) {
            const int b = (int)(gr / p.Mrows), ml = (int)(gr - (long)b * p.Mrows);
            const int oy = ml / p.aW, ox = ml - oy * p.aW;
            const size_t off = ((size_t)(b * p.out.Hs + oy * 4 + (c >> 1)) * p.out.Ws + ox * 4 + (c & 1) * 2) * 4;
            w2x_store_out((half8*)(Og + off), *(const half8*)(Ot + rr * LDO + c * 8));
        }
    }
}

// The slab variant of merge_kernel below, which cunet's down convolutions (64 -> 64, 128 -> 128, LeakyReLU, the input optionally gated by its
// squeeze-excite block) keep: with the gates' table reads in the unrolled nest the direct-fragment kernel needs 168 / 256 registers there
// against this one's 110 / 158 (4 / 3 waves per SIMD).  Output pixel (y, x) reads the input pixels (2y+ky, 2x+kx): for each ky one
// contiguous run of 2*Cin halves, so K = 4*Cin is walked in sub-chunks of SUB contiguous channels.  A wave owns 32 output
// pixels and all N output columns (accumulators stay in registers over the whole K), rows of a sub-chunk pass through the
// wave's LDS slab into A fragments, weights stream through LDS in stages of two n-tiles shared by the four waves.
template <int CIN, int N, int SUB>
struct MergeSlabCfg {
    static constexpr int K = 4 * CIN, NT = N / 16, KST = K / 32, NQ = K / SUB, QPK = 2 * CIN / SUB, KSS = SUB / 32;   // sub-chunks total / per ky, k-steps per sub-chunk
    static constexpr int TT = 2, RW = 32, G = 2, NF = G * KSS, NFW = NF / 4, LDS_ROW = (SUB > N ? SUB : N) + 8, PPC = SUB / 8, NPI = RW * PPC / 64;
    static constexpr int PPO = N / 8, NPO = RW * PPO / 64;
    static constexpr bool GATES = CIN == 64 || CIN == 128;     // cunet's shapes (pixgemm_supported)
    static constexpr int WBUF = NF * 1024, SLAB = RW * LDS_ROW * 2 + RW * 8, SMEM = 2 * WBUF + 4 * SLAB;
    static_assert(NF % 4 == 0 && NT % G == 0 && (2 * CIN) % SUB == 0 && RW * PPC % 64 == 0 && RW * PPO % 64 == 0, "tiling");
};

template <int CIN, int N, int SUB>
__global__ __launch_bounds__(256, 2) void merge_slab_kernel(const GemmParams p) {
    using C = MergeSlabCfg<CIN, N, SUB>;
    constexpr int NT = C::NT, KST = C::KST, NQ = C::NQ, QPK = C::QPK, KSS = C::KSS;
    constexpr int TT = C::TT, RW = C::RW, G = C::G, NFW = C::NFW, LDS_ROW = C::LDS_ROW, PPC = C::PPC, NPI = C::NPI;
    constexpr int WBUF = C::WBUF, SLAB = C::SLAB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* WB = (_Float16*)smem;
    _Float16* Sl = (_Float16*)(smem + 2 * WBUF + wv * SLAB);              // [RW][LDS_ROW]: input sub-chunk, later the output tile
    int* Tb = (int*)(smem + 2 * WBUF + wv * SLAB + RW * LDS_ROW * 2);      // [RW] element offset of input pixel (2y, 2x), -1: no row
    int* Gb = Tb + RW;                                                     // [RW] offset of the row's image in the gate table (b * CIN)
    const long M = (long)p.B * p.Mrows;
    const long row0 = ((long)blockIdx.x * 4 + wv) * RW;
    const long nrows = M - row0 < RW ? M - row0 : RW;
    const _Float16* __restrict__ Xg = (const _Float16*)p.a.p;
    const _Float16* __restrict__ Wf = (const _Float16*)p.wt_frag + lane * 8;   // [NT][KST][64][8]
    auto frag_src = [&](int stage, int f) {   // stage = q * (NT / G) + s; fragment f = t * KSS + ks  ->  n-tile s*G + t, k-step q*KSS + ks
        const int q = stage / (NT / G), s2 = stage - q * (NT / G), t = f / KSS, ks = f - t * KSS;
        return Wf + (size_t)((s2 * G + t) * KST + q * KSS + ks) * 512;
    };
    half8 stg[NFW];
#pragma unroll
    for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(0, wv * NFW + i);
    if (lane < RW) {
        const long gr = row0 + lane;
        int off = -1, gb = 0;
        if (gr < M) {
            const int b = (int)(gr / p.Mrows), ml = (int)(gr - (long)b * p.Mrows);
            const int oy = ml / p.aW, ox = ml - oy * p.aW;
            off = ((b * p.a.Hs + oy * 2 + p.a.y0) * p.a.Ws + ox * 2 + p.a.x0) * CIN;
            gb = b * CIN;
        }
        Tb[lane] = off; Gb[lane] = gb;
    }
#pragma unroll
    for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
    __syncthreads();

    float4v acc[TT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float b = p.bias[t * 16 + fr];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt][t] = (float4v){b, b, b, b};
    }
    constexpr int NSTAGE = NQ * (NT / G);
    int stage = 0;
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
        // rows of this sub-chunk: SUB contiguous halves per output pixel
        const int ky = q / QPK, inner = (q - ky * QPK) * SUB;
        const int shift = ky * p.a.Ws * CIN + inner;
        {
            half8 xr[NPI];
#pragma unroll
            for (int k = 0; k < NPI; ++k) {
                const int idx = k * 64 + lane, rr = idx / PPC, c = idx - rr * PPC;
                const int off = Tb[rr];
                half8 h = {};
                if (off >= 0) {
                    h = *(const half8*)(Xg + (size_t)off + shift + c * 8);
                    if (C::GATES && p.a_scale) h = gate::gate8(h, p.a_scale + Gb[rr] + (inner + c * 8) % CIN);   // squeeze-excite gate of the input map
                }
                xr[k] = h;
            }
#pragma unroll
            for (int k = 0; k < NPI; ++k) {
                const int idx = k * 64 + lane, rr = idx / PPC, c = idx - rr * PPC;
                *(half8*)(Sl + rr * LDS_ROW + c * 8) = xr[k];
            }
        }
        W2X_PHASE_FENCE();
        half8 xa[TT][KSS];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int ks = 0; ks < KSS; ++ks) xa[tt][ks] = *(const half8*)(Sl + (tt * 16 + fr) * LDS_ROW + ks * 32 + g * 8);
        W2X_PHASE_FENCE();
#pragma unroll
        for (int s2 = 0; s2 < NT / G; ++s2, ++stage) {
            const _Float16* wcur = WB + (size_t)(stage & 1) * (WBUF / 2) + lane * 8;
            if (stage + 1 < NSTAGE) {
#pragma unroll
                for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(stage + 1, wv * NFW + i);   // (pinning these at the top of the stage as in
                                                                                                               // pixgemm_kernel measured 3-7 % SLOWER here)
            }
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
                for (int ks = 0; ks < KSS; ++ks) {
                    const half8 wb = *(const half8*)(wcur + (size_t)(t * KSS + ks) * 512);
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) acc[tt][s2 * G + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[tt][ks], wb, acc[tt][s2 * G + t], 0, 0, 0);
                }
            if (stage + 1 < NSTAGE) {
#pragma unroll
                for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)((stage + 1) & 1) * (WBUF / 2) + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
            }
            __syncthreads();
        }
    }
    // ---- output tile through the slab, flat 16-byte stores (rows of the output are contiguous)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[tt][t][j];
                if (p.act == 1) v = v > 0.f ? v : v * p.alpha;
                Sl[(tt * 16 + g * 4 + j) * LDS_ROW + t * 16 + fr] = (_Float16)v;
            }
    W2X_PHASE_FENCE();
    _Float16* __restrict__ Og = (_Float16*)p.out.p + row0 * N;
    const int npieces = nrows > 0 ? (int)nrows * C::PPO : 0;
#pragma unroll
    for (int k = 0; k < C::NPO; ++k) {
        const int idx = k * 64 + lane, rr = idx / C::PPO, c = idx - rr * C::PPO;
        if (idx < npieces) w2x_store_out((half8*)(Og + (size_t)idx * 8), *(const half8*)(Sl + rr * LDS_ROW + c * 8));
    }
}

// Conv 2x2 stride 2: the Swin patch merge (Cin 96 / 192 -> 192) and cunet's down convolutions (64 -> 64, 128 -> 128, LeakyReLU, the
// input optionally gated by its squeeze-excite block).  Output pixel (y, x) reads the input pixels (2y+ky, 2x+kx): for each ky one
// contiguous run of 2*Cin halves, so K = 4*Cin is walked in sub-chunks of SUB contiguous channels.  A wave owns 32 output
// pixels and all N output columns (accumulators stay in registers over the whole K); weights stream through LDS in stages of two
// n-tiles shared by the four waves, requested W2X_MERGE_AHEAD stages ahead.
// The A fragments of a sub-chunk are loaded STRAIGHT from the map (lane (fr, g) of k-step ks takes the 16 bytes at channel 32 ks + 8 g of
// its row: four lanes cover 64 contiguous bytes of a row, the next k-step the other half of the 128-byte line) into one of two
// register sets, and the next sub-chunk's are requested when this one's first stage starts.  Round 2 / early round 3 passed the rows
// through registers and the wave's LDS slab first (48 more registers - no room to request anything ahead, ten registers spilled at
// Cin = 192) and waited for every sub-chunk's rows with nothing else to do: 2 - 4 exposed round trips to memory per workgroup.
// The loop nest is fully unrolled (at most 24 stages), so which register set a stage uses and which loads are in flight at each wait
// are compile-time facts.
#ifndef W2X_MERGE_TT
#define W2X_MERGE_TT 2        // m-tiles of 16 output pixels per wave (1: 64 pixels per workgroup, three workgroups per CU, twice the weight traffic from L2)
#endif
#ifndef W2X_MERGE_AHEAD
#define W2X_MERGE_AHEAD 2
#endif
template <int CIN, int N, int SUB>
struct MergeCfg {
    static constexpr int K = 4 * CIN, NT = N / 16, KST = K / 32, NQ = K / SUB, QPK = 2 * CIN / SUB, KSS = SUB / 32;   // sub-chunks total / per ky, k-steps per sub-chunk
    static constexpr int TT = W2X_MERGE_TT, RW = 16 * TT, G = 2, NF = G * KSS, NFW = NF / 4, LDS_ROW = N + 8;
    static constexpr int PPO = N / 8, NPO = RW * PPO / 64;
    static constexpr int SPQ = NT / G, NSTAGE = NQ * SPQ;      // weight stages per sub-chunk / in all
    static constexpr int AHEAD = W2X_MERGE_AHEAD;
    static constexpr bool GATES = CIN == 64 || CIN == 128;     // cunet's shapes (pixgemm_supported)
    static constexpr int WBUF = NF * 1024, SLAB = RW * LDS_ROW * 2, SMEM = 2 * WBUF + 4 * SLAB;
    static_assert(NF % 4 == 0 && NT % G == 0 && (2 * CIN) % SUB == 0 && RW * PPO % 64 == 0, "tiling");
};

template <int CIN, int N, int SUB>
__global__ __launch_bounds__(256, W2X_MERGE_TT == 1 ? 3 : 2) void merge_kernel(const GemmParams p) {
    using C = MergeCfg<CIN, N, SUB>;
    constexpr int NT = C::NT, KST = C::KST, NQ = C::NQ, QPK = C::QPK, KSS = C::KSS, SPQ = C::SPQ, NSTAGE = C::NSTAGE, D = C::AHEAD;
    constexpr int TT = C::TT, RW = C::RW, G = C::G, NFW = C::NFW, LDS_ROW = C::LDS_ROW;
    constexpr int WBUF = C::WBUF, SLAB = C::SLAB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* WB = (_Float16*)smem;
    _Float16* Sl = (_Float16*)(smem + 2 * WBUF + wv * SLAB);              // [RW][LDS_ROW]: the output tile
    const long M = (long)p.B * p.Mrows;
    const long row0 = ((long)blockIdx.x * 4 + wv) * RW;
    const long nrows = M - row0 < RW ? M - row0 : RW;
    const _Float16* __restrict__ Wf = (const _Float16*)p.wt_frag + lane * 8;   // [NT][KST][64][8]
    auto frag_src = [&](int stage, int f) {   // stage = q * SPQ + s; fragment f = t * KSS + ks  ->  n-tile s*G + t, k-step q*KSS + ks
        const int q = stage / SPQ, s2 = stage - q * SPQ, t = f / KSS, ks = f - t * KSS;
        return Wf + (size_t)((s2 * G + t) * KST + q * KSS + ks) * 512;
    };
    // the map goes through a buffer resource (32-bit byte offsets, pixgemm_supported refuses 4 GB and more): a row that does not exist is
    // offset 0xFFFFFFFF (saturating adds keep it there) and reads zeros
    const __amdgpu_buffer_rsrc_t XB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a.p), 0, (unsigned)((size_t)p.B * p.a.Hs * p.a.Ws * CIN * 2), 0x00020000);

    float bias_r[NT];                              // requested first: the oldest loads, so no later wait is held up by them
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_r[t] = p.bias[t * 16 + fr];
    half8 stg[D][NFW];
#pragma unroll
    for (int i = 0; i < NFW; ++i) stg[0][i] = *(const half8*)frag_src(0, wv * NFW + i);
    unsigned rbase[TT]; int gb[TT];                // this lane's rows tt * 16 + fr: byte offset of channel 8 g of input pixel (2y, 2x); gate table row
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const long gr = row0 + tt * 16 + fr;
        rbase[tt] = 0xFFFFFFFFu; gb[tt] = 0;
        if (gr < M) {
            const int b = (int)(gr / p.Mrows), ml = (int)(gr - (long)b * p.Mrows);
            const int oy = ml / p.aW, ox = ml - oy * p.aW;
            rbase[tt] = (unsigned)((((size_t)(b * p.a.Hs + oy * 2 + p.a.y0) * p.a.Ws + ox * 2 + p.a.x0) * CIN + g * 8) * 2);
            gb[tt] = b * CIN;
        }
    }
    half8 xa[2][TT][KSS];                          // A fragments of sub-chunk q in set q & 1
    auto xload = [&](int q) {
        const int ky = q / QPK, inner = (q - ky * QPK) * SUB;
        const unsigned shift = (unsigned)((ky * p.a.Ws * CIN + inner) * 2);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int ks = 0; ks < KSS; ++ks)
                xa[q & 1][tt][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(XB, __builtin_elementwise_add_sat(rbase[tt], shift + ks * 64u), 0, 0));
    };
    xload(0);
#pragma unroll
    for (int d = 1; d < D; ++d)
        if (d < NSTAGE) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) stg[d][i] = *(const half8*)frag_src(d, wv * NFW + i);
        }
#pragma unroll
    for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[0][i];
    float4v acc[TT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt][t] = (float4v){bias_r[t], bias_r[t], bias_r[t], bias_r[t]};
    __syncthreads();

#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int s2 = 0; s2 < SPQ; ++s2) {
            const int stage = q * SPQ + s2;        // a constant after unrolling
            const _Float16* wcur = WB + (size_t)(stage & 1) * (WBUF / 2) + lane * 8;
            if (stage + D < NSTAGE) {              // set stage % D carried this stage, which reached LDS at the end of the previous one
#pragma unroll
                for (int i = 0; i < NFW; ++i) stg[stage % D][i] = *(const half8*)frag_src(stage + D, wv * NFW + i);
            }
            if (s2 == 0 && q + 1 < NQ) xload(q + 1);
            W2X_PHASE_FENCE();                     // the requests stay at the top of the stage
            if (C::GATES && s2 == 0 && p.a_scale) {   // squeeze-excite gate of the input map: fp16(x * s), the rounding of the in-place pass
                const int ky = q / QPK, inner = (q - ky * QPK) * SUB;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int ks = 0; ks < KSS; ++ks) xa[q & 1][tt][ks] = gate::gate8(xa[q & 1][tt][ks], p.a_scale + gb[tt] + (inner + ks * 32 + g * 8) % CIN);
            }
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
                for (int ks = 0; ks < KSS; ++ks) {
                    const half8 wb = *(const half8*)(wcur + (size_t)(t * KSS + ks) * 512);
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) acc[tt][s2 * G + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[q & 1][tt][ks], wb, acc[tt][s2 * G + t], 0, 0, 0);
                }
            if (stage + 1 < NSTAGE) {
#pragma unroll
                for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)((stage + 1) & 1) * (WBUF / 2) + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[(stage + 1) % D][i];
            }
            __syncthreads();
        }
    }
    // ---- output tile through the slab, flat 16-byte stores (rows of the output are contiguous)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[tt][t][j];
                if (p.act == 1) v = v > 0.f ? v : v * p.alpha;
                Sl[(tt * 16 + g * 4 + j) * LDS_ROW + t * 16 + fr] = (_Float16)v;
            }
    W2X_PHASE_FENCE();
    _Float16* __restrict__ Og = (_Float16*)p.out.p + row0 * N;
    const int npieces = nrows > 0 ? (int)nrows * C::PPO : 0;
#pragma unroll
    for (int k = 0; k < C::NPO; ++k) {
        const int idx = k * 64 + lane, rr = idx / C::PPO, c = idx - rr * C::PPO;
        if (idx < npieces) w2x_store_out((half8*)(Og + (size_t)idx * 8), *(const half8*)(Sl + rr * LDS_ROW + c * 8));
    }
}

template <int CIN, int N, int SUB>
hipError_t launch_merge(const GemmParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    const long M = (long)p.B * p.Mrows;
    const dim3 grid((unsigned)((M + 127) / 128));
    if constexpr (CIN == 64 || CIN == 128) {                       // cunet's shapes
        constexpr int SM = MergeSlabCfg<CIN, N, SUB>::SMEM;
        if (hipError_t e = ensure_dynamic_lds((const void*)merge_slab_kernel<CIN, N, SUB>, SM, lds_ok); e != hipSuccess) return e;
        hipLaunchKernelGGL((merge_slab_kernel<CIN, N, SUB>), grid, dim3(256), SM, s, p);
    } else {
        constexpr int SM = MergeCfg<CIN, N, SUB>::SMEM, BMQ = 4 * MergeCfg<CIN, N, SUB>::RW;
        if (hipError_t e = ensure_dynamic_lds((const void*)merge_kernel<CIN, N, SUB>, SM, lds_ok); e != hipSuccess) return e;
        hipLaunchKernelGGL((merge_kernel<CIN, N, SUB>), dim3((unsigned)((M + BMQ - 1) / BMQ)), dim3(256), SM, s, p);
    }
    return hipGetLastError();
}

}  // namespace

// true if this launch can take the streaming kernel (everything else stays on gemm_kernel)
bool pixgemm_supported(const GemmParams& p) {
    const bool off = switches().no_pixgemm;   // reference path (switches.h)
    if ((p.a_scale || p.res_scale) && p.a.Cs != 64 && p.a.Cs != 128) return false;   // gated operands: compiled into cunet's shapes only (PixCfg / MergeCfg GATES)
    if (!off && p.wt_frag && p.amode == 2 && p.kh == 2 && p.kw == 2 && p.stride == 2 && p.omode == 0 && !p.ln && (p.act == 0 || p.act == 1) && !p.has_clip &&
        !p.stats_out && !p.pool_out && !p.res.p && !p.res2.p && p.out.Cs == p.N && p.Kw == p.K && p.K == 4 * p.a.Cs && (long)p.out.Hs * p.out.Ws == p.Mrows && p.out.Ws == p.aW &&
        (((p.a.Cs == 96 || p.a.Cs == 192) && p.N == 192 && (size_t)p.B * p.a.Hs * p.a.Ws * p.a.Cs * 2 < 0xFFFF0000u) ||   // (merge_kernel reads its map through 32-bit offsets)
         (p.a.Cs == 64 && p.N == 64) || (p.a.Cs == 128 && p.N == 128))) return true;   // patch merge / cunet down convolution
    // rows = a Linear, or a 1x1 convolution: cunet's 2x2 stride-2 ConvTranspose is lowered to 1x1 + pixel shuffle with LeakyReLU and a
    // cropped skip add (K = 64 / 128)
    const bool rows = p.amode == 0 || (p.amode == 2 && p.kh == 1 && p.kw == 1);
    if (off || !p.wt_frag || p.omode != 2 || !rows || p.ln || (p.act != 0 && p.act != 1) || p.stats_out || p.pool_out || p.res2.p) return false;
    if (p.stride != 1 || p.a.y0 || p.a.x0 || p.a.Ws != p.aW || (long)p.a.Hs * p.a.Ws != p.Mrows || p.a.Cs != p.K || p.Kw != p.K) return false;
    if (p.N != p.r * p.r * p.out.Cs) return false;
    if (p.K == 96 && p.out.Cs == 4 && p.r == 4 && p.N == 64 && !p.res.p && p.act == 0 && p.amode == 0 && !p.a_scale) return true;   // image head
    if (p.has_clip || p.Cout != p.out.Cs || (p.res.p && p.res.Cs != p.out.Cs)) return false;
    {   // pixgemm_kernel addresses its three maps with 32-bit byte offsets and keeps the bias of at most 16 sub-pixels in LDS
        const size_t lim = 0xFFFF0000u;
        if (p.r > 4 || (size_t)p.B * p.Mrows * p.K * 2 >= lim || (size_t)p.B * p.out.Hs * p.out.Ws * p.out.Cs * 2 >= lim ||
            (p.res.p && (size_t)p.B * p.res.Hs * p.res.Ws * p.res.Cs * 2 >= lim)) return false;
    }
    if ((p.K == 64 && p.out.Cs == 64) || (p.K == 128 && p.out.Cs == 128)) return true;
    return (p.K == 192 && (p.out.Cs == 96 || p.out.Cs == 192));
}

hipError_t launch_pixgemm(const GemmParams& p, hipStream_t s) {
    if (p.amode == 2 && p.kh == 2) return p.a.Cs == 96 ? launch_merge<96, 192, 192>(p, s) : p.a.Cs == 192 ? launch_merge<192, 192, 192>(p, s) : p.a.Cs == 64 ? launch_merge<64, 64, 128>(p, s) : launch_merge<128, 128, 128>(p, s);
    if (p.out.Cs == 4) {
        constexpr int SM = 4 * 64 * 104 * 2;
        const long M = (long)p.B * p.Mrows;
        hipLaunchKernelGGL(toimage_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), SM, s, p);
        return hipGetLastError();
    }
    if (p.K == 192 && p.out.Cs == 96) return launch_pix<192, 96, 2>(p, s);
    if (p.K == 192 && p.out.Cs == 192) return launch_pix<192, 192, 2, W2X_PIX192_TT>(p, s);
    if (p.K == 64 && p.out.Cs == 64) return launch_pix<64, 64, 2>(p, s);
    if (p.K == 128 && p.out.Cs == 128) return launch_pix<128, 128, 2, 1>(p, s);   // (round 6: one m-tile per wave, three workgroups per CU: 0.46 -> 0.41 ms per config-2 frame; the frame itself +-0.1 %, the launch runs under the other tile group's convolutions)
    return hipErrorInvalidValue;
}

}  // namespace w2x
