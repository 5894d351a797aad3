// Direct 3x3 convolution (stride 1, valid) onto FEW output channels for gfx950 - cunet's two image heads:
//   * unet2 conv_bottom: 64 -> 3 channels (stored as 4 halves per pixel), cropped skip add from unet1, clip to [0, 1];
//   * unet1 conv_bottom: the 4x4 stride-2 ConvTranspose 64 -> 3, lowered to a 3x3 convolution onto 4 sub-pixels x 4 stored channels
//     with a 2x pixel shuffle.
// gemm_kernel (k_gemm.hip) runs these as an implicit GEMM that fetches every input pixel nine times and pads N to a 64-row tile:
// 1.30 + 0.39 ms of config 2's 11.5 ms at 0.9 / 1.1 TB/s (now 0.57 + 0.22).  The launches are pure input streams (1.19 GB + 0.35 GB read, 16 MFMA per
// 16 pixels), so the kernel is k_conv3.hip's halo tile with everything else removed:
//   * a workgroup owns 4 output rows x 64 columns; per 32-channel chunk its 6 x 66 pixel halo tile goes to LDS in one batch of
//     16-byte loads per thread (25 KB, unpadded, pieces rotated so that fragment reads and the stores are free of bank conflicts:
//     halo_slot), four workgroups per CU - the launch is fetch -> products -> store in every workgroup, so it lives off workgroups
//     that are out of phase (a single 57 KB tile with all 64 channels at two workgroups per CU ran the 9.3 M-pixel head in 0.69 ms,
//     this runs it in 0.57: 2.1 TB/s of input, 3.2 TB/s into the CUs with the 1.55x halo overlap of 4-row tiles);
//   * the product is transposed (out^T = W X^T): the weights are the A operand - 9 taps per chunk, loaded per wave straight from
//     the [N][K] matrix (rows >= N read as zero) into registers - the pixels the B operand, so a lane ends up with 4 consecutive
//     stored channels of ONE pixel: channels 4g .. 4g+3 of pixel fr.  Rows output: lanes g = 0 hold the pixel (8-byte store);
//     pixel-shuffle output: g is the sub-pixel (dy, dx) = (g >> 1, g & 1), every lane stores 8 bytes;
//   * epilogue in registers: bias (initial accumulator), LeakyReLU / none, skip add, clip, ONE rounding to fp16 (gemm_kernel rounds
//     before and after the skip add).
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

struct Conv3hCfg {
    static constexpr int TH = 4, TW = 64, HR = TH + 2, HC = TW + 2;
    static constexpr int ROWB = HC * 64;                               // bytes per halo row (one chunk of 32 channels)
    static constexpr int NP = HR * HC * 4, NI = (NP + 255) / 256;      // 16-byte pieces of a chunk's halo tile, per thread
    static constexpr int SMEM = HR * ROWB;
};
// byte offset of piece pc (8 channels) of halo pixel x inside its row: 64 bytes per pixel, the pieces rotated by 2 * ((x >> 2) & 3)
// slots - ds_read_b128 serves 16-lane groups that pair k-groups over complementary row sets, the rotation keeps them conflict-free
// (k_conv3.hip has the derivation)
__device__ __forceinline__ int halo_slot(int x, int pc) { return x * 64 + ((pc + 2 * ((x >> 2) & 3)) & 3) * 16; }

template <int PIX, bool TWO>
__global__ __launch_bounds__(256, 4) void conv3h_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y) {
    using C = Conv3hCfg;
    constexpr int HR = C::HR, HC = C::HC;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const int tpi = tiles_x * tiles_y;
    const int b = blockIdx.x / tpi, trem = blockIdx.x - b * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int Cin = p.a.Cs, nchunk = Cin / 32;
    const _Float16* __restrict__ Ag = (const _Float16*)p.a.p + ((size_t)(b * p.a.Hs + p.a.y0 + oy0) * p.a.Ws + p.a.x0 + ox0) * Cin;

    // halo tile of a 32-channel chunk: pixels past the input extent of a ragged tile (input extent = output extent + 2) are zeros
    const int hrows = min(HR, Ho + 2 - oy0), hcols = min(HC, Wo + 2 - ox0);
    const _Float16* __restrict__ Wt = (const _Float16*)p.wt + (size_t)fr * p.Kw + g * 8;   // [N][Kw], k = tap * Cin + channel
    float4v bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = 4 * g + j < p.N ? p.bias[4 * g + j] : 0.f;
    float4v acc[4] = {bv, bv, bv, bv};                                 // wave wv: output row oy0 + wv, four groups of 16 pixels
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = halo_slot(kx + fr, g);
    const unsigned char* xrow = smem + wv * C::ROWB;
    auto fetch = [&](int kc, half8 (&h)[C::NI]) {
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const int idx = tid + i * 256, pix = idx >> 2, c8 = idx & 3, hr = pix / HC, hc = pix - hr * HC;
            h[i] = (half8){};
            if (idx < C::NP && hr < hrows && hc < hcols) h[i] = *(const half8*)(Ag + ((size_t)hr * p.a.Ws + hc) * Cin + kc * 32 + c8 * 8);
        }
    };
    auto chunk = [&](int kc, const half8 (&h)[C::NI]) {
        // weights of this chunk: fragment (tap) of lane (fr, g) = W[fr][tap][32 kc + 8g .. + 7], rows >= N read as zero
        half8 wf[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wf[t] = fr < p.N ? *(const half8*)(Wt + t * Cin + kc * 32) : (half8){};
        __syncthreads();                                   // the previous chunk's products are done with the halo tile
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const int idx = tid + i * 256, pix = idx >> 2, c8 = idx & 3, hr = pix / HC, hc = pix - hr * HC;
            if (idx < C::NP) *(half8*)(smem + hr * C::ROWB + halo_slot(hc, c8)) = h[i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            half8 xa[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(xrow + ky * C::ROWB + xoff[kx] + mt * 1024);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], xa[mt], acc[mt], 0, 0, 0);
        }
    };
    if (TWO) {
        // Both 32-channel chunks of a 64-channel map are requested before anything waits (late round 4): the second chunk's halo tile travels under the first
        // chunk's LDS stores, barriers and products instead of being asked for after them - one exposed fetch per workgroup instead of two; 28 more registers.
        half8 h0[C::NI], h1[C::NI];
        fetch(0, h0);
        fetch(1, h1);
        chunk(0, h0);
        chunk(1, h1);
    } else {
#pragma unroll 1
        for (int kc = 0; kc < nchunk; ++kc) {
            half8 h[C::NI];
            fetch(kc, h);
            chunk(kc, h);
        }
    }

    const int oy = oy0 + wv;
    if (oy >= Ho) return;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int ox = ox0 + mt * 16 + fr;
        if (ox >= Wo || (!PIX && g != 0)) continue;
        const int Y = PIX ? 2 * oy + (g >> 1) : oy, X = PIX ? 2 * ox + (g & 1) : ox;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = acc[mt][j]; if (p.act == 1) v[j] = v[j] > 0.f ? v[j] : v[j] * p.alpha; }
        if (!PIX && p.res.p) {
            const half4 r = *(const half4*)((const _Float16*)p.res.p + ((size_t)(b * p.res.Hs + Y + p.res.y0) * p.res.Ws + X + p.res.x0) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
        if (p.has_clip) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], p.clip_lo), p.clip_hi);
        }
        *(half4*)((_Float16*)p.out.p + ((size_t)(b * p.out.Hs + Y) * p.out.Ws + X) * 4) = (half4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
}

template <int PIX, bool TWO>
hipError_t launch_c3h2(const GemmParams& p, int Ho, int Wo, hipStream_t s) {
    using C = Conv3hCfg;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3h_kernel<PIX, TWO>, C::SMEM, lds_ok); e != hipSuccess) return e;
    const int tiles_x = (Wo + C::TW - 1) / C::TW, tiles_y = (Ho + C::TH - 1) / C::TH;
    hipLaunchKernelGGL((conv3h_kernel<PIX, TWO>), dim3((unsigned)(p.B * tiles_x * tiles_y)), dim3(256), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y);
    return hipGetLastError();
}
template <int PIX>
hipError_t launch_c3h(const GemmParams& p, int Ho, int Wo, hipStream_t s) {
    return p.a.Cs == 64 ? launch_c3h2<PIX, true>(p, Ho, Wo, s) : launch_c3h2<PIX, false>(p, Ho, Wo, s);
}


// Round 6: the same convolution as a WALK down a column strip, for the 64-channel maps both heads read.  conv3h_kernel above fetches a 6 x 66 halo tile per 4 output rows
// (every input row 1.5 times, in 64-byte half-pixel pieces) and each workgroup does fetch -> wait -> products -> store once: 2.3 TB/s on config 2's 1.2 GB head.  Here a
// workgroup owns 64 output columns and RB output rows and walks them four rows (one per wave) at a time:
//   * the input rows live in a ring of eight row slots in LDS (both 32-channel chunks of a row side by side, k_conv3.hip's rotated layout: 66 KB, two workgroups per CU);
//     a step needs rows 4s .. 4s + 5, of which four are new - every input row is fetched ONCE per strip (plus two per row block), as whole 128-byte pixels, consecutive
//     threads on consecutive 16-byte pieces;
//   * the four rows of step s + 2 are requested before the products of step s (nine 16-byte loads per thread; two batches wait in registers), the rows of step s + 1 are
//     stored into the slots of rows the products no longer need behind one barrier - memory is always in flight; the skip pixels of a step's outputs are requested first;
//   * the weights of all 18 (chunk, tap) fragments stay in registers for the whole walk (72 registers), the products run in conv3h_kernel's order (chunk 0 taps 0 .. 8,
//     chunk 1 taps 0 .. 8, bias as the initial accumulator) and the epilogue is its epilogue: the same bytes (test: the debug switch no_conv3h_walk).
struct Conv3wCfg {
    static constexpr int TW = 64, HC = TW + 2, NSLOT = 8;
    static constexpr int ROWB = HC * 64, SLOTB = 2 * ROWB;             // a chunk's row, a row slot (two chunks)
    static constexpr int NPR = HC * 8;                                 // 16-byte pieces of an input row (64 channels)
    static constexpr int NI4 = (4 * NPR + 255) / 256, NI2 = (2 * NPR + 255) / 256;   // per thread: a batch of four rows, of two rows
    static constexpr int SMEM = NSLOT * SLOTB;
};

template <int PIX>
__global__ __launch_bounds__(256, 2) void conv3h_walk_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int nby, int RB) {
    using C = Conv3wCfg;
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    constexpr int HC = C::HC;
    constexpr unsigned kNoPix = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const int per_img = tiles_x * nby;
    const int b = blockIdx.x / per_img, trem = blockIdx.x - b * per_img;
    const int by = trem / tiles_x, tx = trem - by * tiles_x;
    const int oy_base = by * RB, ox0 = tx * C::TW;
    const int rows_here = min(RB, Ho - oy_base), nsteps = (rows_here + 3) >> 2;
    const int in_rows = min(rows_here + 2, Ho + 2 - oy_base);           // input rows of this block that exist (the rest are zeros)
    const int hcols = min(HC, Wo + 2 - ox0);
    constexpr int Cin = 64;
    const unsigned rowb = (unsigned)p.a.Ws * Cin * 2u;
    const __amdgpu_buffer_rsrc_t A = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const _Float16*)p.a.p + ((size_t)(b * p.a.Hs + p.a.y0 + oy_base) * p.a.Ws + p.a.x0 + ox0) * Cin), 0, 0x7FFFFFFFu, 0x00020000);

    // weights: fragment (chunk, tap) of lane (fr, g) = W[fr][tap][32 chunk + 8g .. + 7], rows >= N read as zero
    const _Float16* __restrict__ Wt = (const _Float16*)p.wt + (size_t)fr * p.Kw + g * 8;   // [N][Kw], k = tap * Cin + channel
    half8 wf[2][9];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int t = 0; t < 9; ++t) wf[kc][t] = fr < p.N ? *(const half8*)(Wt + t * Cin + kc * 32) : (half8){};
    float4v bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = 4 * g + j < p.N ? p.bias[4 * g + j] : 0.f;
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = halo_slot(kx + fr, g);

    // a batch of rows: piece idx = tid + 256 i -> (row of the batch, pixel, 16-byte piece of the pixel's 128 bytes)
    auto piece = [&](int i, int& r, int& px, int& c) { const int idx = tid + i * 256; r = idx / C::NPR; const int rem = idx - r * C::NPR; px = rem >> 3; c = rem & 7; };
    auto request = [&](int row0, int nrow, int i) -> uint4v {            // piece i of the batch of nrow rows that starts at input row row0 (block-relative)
        int r, px, c;
        piece(i, r, px, c);
        const bool in = r < nrow && row0 + r < in_rows && px < hcols;
        return __builtin_amdgcn_raw_buffer_load_b128(A, in ? (unsigned)(r * rowb) + (unsigned)(px * 128 + c * 16) : kNoPix, (unsigned)row0 * rowb, 0);
    };
    auto deposit = [&](int row0, int nrow, int i, uint4v v) {
        int r, px, c;
        piece(i, r, px, c);
        if (r < nrow) *(uint4v*)(smem + ((row0 + r) & (C::NSLOT - 1)) * C::SLOTB + (c >> 2) * C::ROWB + halo_slot(px, c & 3)) = v;
    };
    {   // rows 0 .. 5 of the block
        uint4v h2[C::NI2], h4[C::NI4];
#pragma unroll
        for (int i = 0; i < C::NI2; ++i) h2[i] = request(0, 2, i);
#pragma unroll
        for (int i = 0; i < C::NI4; ++i) h4[i] = request(2, 4, i);
#pragma unroll
        for (int i = 0; i < C::NI2; ++i) deposit(0, 2, i, h2[i]);
#pragma unroll
        for (int i = 0; i < C::NI4; ++i) deposit(2, 4, i, h4[i]);
    }
    __syncthreads();
    // Two batches travel: the rows of step s + 1 (requested during step s - 1, stored behind step s's products) and those of step s + 2 (requested at the top of step s).
    // The ring takes rows one step ahead only - the second batch waits in registers.
    uint4v ha[C::NI4], hb[C::NI4];
    const int res_on = !PIX && p.res.p != nullptr;
    if (nsteps > 1) {
#pragma unroll
        for (int i = 0; i < C::NI4; ++i) ha[i] = request(6, 4, i);      // rows of step 1
    }
    auto step = [&](const int s, uint4v (&nxt)[C::NI4], uint4v (&far)[C::NI4]) __attribute__((always_inline)) {
        // (nxt: the rows of step s + 1, on their way since step s - 1; far: free, takes the rows of step s + 2)
        const int oy = oy_base + 4 * s + wv;
        half4 rs[4];                                                    // the skip pixels of this step's outputs: requested first, so that the epilogue waits for them only
        if (res_on) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ox = min(ox0 + mt * 16 + fr, Wo - 1), oyc = min(oy, Ho - 1);
                rs[mt] = *(const half4*)((const _Float16*)p.res.p + ((size_t)(b * p.res.Hs + oyc + p.res.y0) * p.res.Ws + ox + p.res.x0) * 4);
            }
            asm volatile("" ::: "memory");
        }
        if (s + 2 < nsteps) {
#pragma unroll
            for (int i = 0; i < C::NI4; ++i) far[i] = request(4 * s + 10, 4, i);
        }
        asm volatile("" ::: "memory");
        float4v acc[4] = {bv, bv, bv, bv};                              // wave wv: output row oy_base + 4s + wv, four groups of 16 pixels
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const unsigned char* xrow = smem + ((4 * s + wv + ky) & (C::NSLOT - 1)) * C::SLOTB + kc * C::ROWB;
                half8 xa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(xrow + xoff[kx] + mt * 1024);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kc][t], xa[mt], acc[mt], 0, 0, 0);
            }
        if (oy < oy_base + rows_here) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ox = ox0 + mt * 16 + fr;
                if (ox >= Wo || (!PIX && g != 0)) continue;
                const int Y = PIX ? 2 * oy + (g >> 1) : oy, X = PIX ? 2 * ox + (g & 1) : ox;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = acc[mt][j]; if (p.act == 1) v[j] = v[j] > 0.f ? v[j] : v[j] * p.alpha; }
                if (res_on) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rs[mt][j];
                }
                if (p.has_clip) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], p.clip_lo), p.clip_hi);
                }
                *(half4*)((_Float16*)p.out.p + ((size_t)(b * p.out.Hs + Y) * p.out.Ws + X) * 4) = (half4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
        if (s + 1 < nsteps) {
            __syncthreads();                                            // every wave is done with rows 4s, 4s + 1, whose slots rows 4s + 8, 4s + 9 take
#pragma unroll
            for (int i = 0; i < C::NI4; ++i) deposit(4 * s + 6, 4, i, nxt[i]);
            __syncthreads();
        }
    };
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 2) {
        step(s, ha, hb);
        if (s + 1 < nsteps) step(s + 1, hb, ha);
    }
}

template <int PIX>
hipError_t launch_c3h_walk(const GemmParams& p, int Ho, int Wo, hipStream_t s) {
    using C = Conv3wCfg;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3h_walk_kernel<PIX>, C::SMEM, lds_ok); e != hipSuccess) return e;
    const int tiles_x = (Wo + C::TW - 1) / C::TW;
    const int strips = p.B * tiles_x;
    // row blocks: about 3 000 workgroups per launch (six rounds of the chip's 512 slots), at least 8 output rows each (a block re-reads two input rows)
    int nby = (3072 + strips - 1) / strips;
    nby = std::max(1, std::min(nby, (Ho + 7) / 8));
    const int RB = ((Ho + nby - 1) / nby + 3) / 4 * 4;
    nby = (Ho + RB - 1) / RB;
    hipLaunchKernelGGL((conv3h_walk_kernel<PIX>), dim3((unsigned)(strips * nby)), dim3(256), C::SMEM, s, p, Ho, Wo, tiles_x, nby, RB);
    return hipGetLastError();
}

}  // namespace

bool conv3h_supported(const GemmParams& p) {
    if (switches().no_conv3h || p.a_scale || p.res_scale || p.amode != 2 || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.ln || (p.act != 0 && p.act != 1) || p.stats_out || p.pool_out || p.res2.p) return false;
    const int Cin = p.a.Cs;
    if (Cin % 32 || Cin > 256 || p.K != 9 * Cin || p.Kw != p.K || p.out.Cs != 4 || p.aW <= 0 || p.Mrows % p.aW || p.B <= 0) return false;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    if (p.a.y0 < 0 || p.a.x0 < 0 || p.a.y0 + Ho + 2 > p.a.Hs || p.a.x0 + Wo + 2 > p.a.Ws) return false;
    if (p.omode == 0) {   // rows: (b, m / aW, m % aW) of the output view, skip add from a cropped view of the same extent
        if (p.N > 4 || p.out.Hs < Ho || p.out.Ws < Wo) return false;
        return !p.res.p || (p.res.Cs == 4 && p.res.y0 >= 0 && p.res.x0 >= 0 && p.res.y0 + Ho <= p.res.Hs && p.res.x0 + Wo <= p.res.Ws);
    }
    return p.omode == 2 && p.r == 2 && p.N == 16 && !p.res.p && p.out.Hs >= 2 * Ho && p.out.Ws >= 2 * Wo;
}

hipError_t launch_conv3h(const GemmParams& p, hipStream_t s) {
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    if (p.a.Cs == 64 && !switches().no_conv3h_walk && (size_t)(Ho + 8) * p.a.Ws * 128 <= 0x7FFFFFFFull)      // the walk (32-bit offsets inside a block's rows)
        return p.omode == 2 ? launch_c3h_walk<1>(p, Ho, Wo, s) : launch_c3h_walk<0>(p, Ho, Wo, s);
    return p.omode == 2 ? launch_c3h<1>(p, Ho, Wo, s) : launch_c3h<0>(p, Ho, Wo, s);
}

}  // namespace w2x
