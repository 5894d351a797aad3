// Direct 3x3 convolution (stride 1, valid) for gfx950 - cunet's 3x3 layers onto 64 / 128 / 256 channels:
//     out[b][y][x][n] = act( sum_{ky,kx,c} in[b][y+ky][x+kx][c] * W[n][(ky*3+kx)*Cin + c] + bias[n] )
// gemm_kernel (k_gemm.hip) treats the same convolution as an implicit GEMM whose A rows are gathered per K-chunk, i.e. every input
// pixel is fetched from L2 nine times (once per tap).  Here a workgroup owns an output tile of 8 rows x 64 columns of one 64-channel
// block; per chunk of 32 input channels its 10 x 66 pixel halo tile goes to LDS once and serves all nine taps.
//   * a wave owns TWO adjacent output rows (2 x 64 pixels x 64 channels = 128 accumulator registers): a halo row's fragments are
//     read once and feed tap ky of the upper and tap ky - 1 of the lower row - 48 instead of 72 fragment reads per chunk;
//   * the weights never enter LDS: they are stored per 64-channel block and k-step as four contiguous fragments (fragorder.h
//     frag_conv3b) and stream from L2 through a ring of three taps (48 registers), each tap fetched one to two steps (32-48 MFMA)
//     before its first use - weight traffic on the vector-memory path, pixel traffic on the LDS path;
//   * barriers only around the halo tile of a chunk (288 MFMA per wave between them), two workgroups per CU;
//   * halo image in LDS: [10 rows][66 pixels][4 pieces of 8 channels], 64 bytes per pixel, no padding, the pieces of a pixel rotated
//     by 2 * ((x >> 2) & 3) slots.  A ds_read_b128 is served in groups of 16 lanes that pair the k-groups (g = 0 with 1, 2 with 3)
//     over complementary row sets ({0-3, 12-15} with {4-11}: MI355X_MICROARCH.md, LDS) on 64 banks; of the four pixels of a group
//     that share x mod 4 two are read for piece g and two for g + 1, and the rotation sends them to four different slots:
//     conflict-free, where a padded pixel-major tile (pixel stride 80 bytes) pays a second cycle on every group
//     (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).  The stores (8 lanes = 2 whole pixels) are conflict-free too;
//   * the product is transposed (out^T = W X^T) with the A rows a permutation of the channels (row 4g + j of n-tile nt = channel
//     32 (nt >> 1) + 8g + 4 (nt & 1) + j of the block): a lane ends up with two runs of 8 consecutive channels of one pixel and a
//     store instruction writes 64 contiguous bytes per pixel, straight from registers; bias is the initial accumulator, LeakyReLU
//     max(v, v * alpha); squeeze-excite pooling partials per workgroup in a fixed order.
// Input, weights and output go through buffer resources (32-bit offsets, rows past a ragged edge read zeros / drop stores).
// (Round 6: the weights DO enter LDS now - WLDS below; the ring remains for the UP mode.  And the ablation figures that follow were taken with wrong data, which at the
// board's power cap measures the clock the garbage frees, not the removed work: DESIGN section 4.)
// Measured on config 2 (tools/op_times.py): the 3x3 layers 6.2 -> 5.4 ms per frame (550-770 TFLOP/s) against the first schedule
// (round-2 git history, tools/ab/k_conv3_v1.hip: one tap of one chunk per barrier, both operands through LDS).  What is left is not the products:
// without the halo fetch the layers run 30-35 % faster, without the stores 25 % - fetch, products and stores of a workgroup run one
// after the other and the two workgroups of a CU stay in phase.  A persistent variant with the halo tiles double-buffered by
// LDS-DMA (round-2 git history, tools/ab/k_conv3c_persistent.hip) overlaps them in program order but came out 2-4 % slower: one 72 KB tile in flight
// per CU leaves the fetch latency-bound (16k cycles per step with the products removed), and LDS has no room for a third buffer.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr unsigned kNoPix = 0xFFFFFFFFu;     // buffer offset past every resource: reads zeros, drops stores
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);   // raw buffer, 32-bit offsets, bounds-checked
}

struct Conv3Cfg {
    static constexpr int TH = 8, TW = 64, HR = TH + 2, HC = TW + 2;
    static constexpr int ROWB = HC * 64;                               // bytes per halo row (32 channels)
    static constexpr int SMEM = HR * ROWB;
    static constexpr int WCHUNK = 9 * 4 * 1024;                         // a chunk's weights: 9 taps x 4 n-tiles of 1 KiB fragments (conv3_kernel WLDS)
};
// byte offset of piece pc (8 channels) of halo pixel x inside its row
__device__ __forceinline__ int halo_slot(int x, int pc) { return x * 64 + ((pc + 2 * ((x >> 2) & 3)) & 3) * 16; }

// STEM (round 6): the launch also computes its own input.  Both U-Nets of cunet open with 3x3 (the 4-halves-per-pixel tile -> 32 channels, k_stem.hip) followed by
// 3x3 32 -> 64; the 32-channel map between them (full resolution: the largest map of the graph per channel) is written by one launch and read by the next and by nothing
// else.  32 channels are exactly ONE chunk of this kernel, so with ps = the stem's parameters the chunk's 10 x 66 halo tile is not fetched but COMPUTED: 42 groups of 16
// halo pixels, the four waves take them round-robin, each group is stem_kernel<2>'s three 16x16x16 k-steps on operands read straight from the input tile (8 bytes per
// pixel: L2-resident), bias as the initial accumulator, LeakyReLU, fp16 - the same instructions on the same operands as stem_kernel, so the tile holds the bytes that
// kernel would have stored (bit-identical frames by test).  A lane ends with channels 8g .. 8g + 7 of one pixel = piece g of the halo layout: one 16-byte LDS store.
// The stem's products are recomputed for the halo (x 1.29), 8 % on top of this launch's matrix work; the stem launch, its stores and this launch's halo loads go.
// Config 2 (tools/ab/switch_bench.py, profiles/r6_kernels/cunet_stem_fold_ab.txt): stem 0.365 + convolution 0.99 ms per frame as two launches, 1.05-1.07 ms as one; frame 7.74-7.82
// against 7.88-7.95 ms.  (With the operands requested in three rounds the launch took 1.15 ms: each round is an L2 round trip in front of the products.)
// UP (round 6): the launch computes its input from the transposed convolution in front of it.  cunet's decoders go  x = LeakyReLU(ConvTranspose 2x2 stride 2 (x * gate)) + skip,
// then 3x3 64 -> 64: as launches, a pixel-shuffle projection (k_pixgemm.hip: 64 -> 4 sub-pixels x 64, gate on its input rows, skip rows added in its epilogue) that
// writes the largest 64-channel map of the graph (1.2 GB per pass at config 2) and this convolution that reads it back.  With ps = the projection's parameters a
// chunk's halo tile is assembled in LDS instead: (1) the halo fetch reads the SKIP map's pixels (same extent, same 64-byte pieces: ps.res in place of p.a), each lane
// the pieces it will add to; (2) the projection's input pixels under the tile - 6 x 34 of them, gated on arrival like pixgemm_kernel's rows - are staged once per workgroup behind the halo
// tile (29 KB, pixel stride 144 bytes); (3) wave w owns sub-pixel class (dy, dx) = (w >> 1, w & 1): its 5 x 33 halo pixels are eleven groups of 16, each two k-steps
// of v_mfma_f32_16x16x32_f16 per 16 channels with the bias as the initial accumulator - pixgemm_kernel's products with the operands' roles exchanged (rows = channels,
// so a lane ends with channels 8g .. 8g + 7 of one pixel = piece g of the halo layout) - LeakyReLU, fp16, and the fp16 add of the skip piece, which was requested before the products and arrives under them; one 16-byte LDS store.
// The projection launch, its 1.2 GB of stores and this launch's 1.2 GB of halo loads go; the projection's products are recomputed for the halo (x 1.29), 14 % on top
// of this launch's matrix work.  Bit-identical to the two launches by test (same products in the same order, same roundings).
template <bool POOL, int MODE>
__global__ __launch_bounds__(256, 2) void conv3_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, int nblk, int xcd_order, const GemmParams ps) {
    using C = Conv3Cfg;
    constexpr bool STEM = MODE == 1, UP = MODE == 2;
    constexpr int HR = C::HR, HC = C::HC;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    // The 2 - 4 output blocks of a pixel tile share its halo.  Workgroups are handed to the eight XCDs round-robin (workgroup q -> XCD q % 8), each XCD
    // with its own L2: as consecutive workgroups the blocks of a tile land on different XCDs and each fetches the halo into its own L2.  XCD-grouped
    // order (round 4): the r-th workgroup of an XCD (r = q / 8) is block r % nblk of tile (r / nblk) * 8 + xcd, so the blocks of a tile are consecutive
    // workgroups of ONE XCD and the halo goes through one L2.
    int nb, tidx;
    const int tpi = tiles_x * tiles_y;
    if (xcd_order) {
        const int q = blockIdx.x, xcd = q & 7, r = q >> 3;
        tidx = (r / nblk) * 8 + xcd; nb = r % nblk;
        if (tidx >= p.B * tpi) return;                                // (the grid is rounded up to whole rounds of eight tiles)
    } else { nb = blockIdx.x % nblk; tidx = blockIdx.x / nblk; }
    const int b = tidx / tpi, trem = tidx - b * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW, n0 = nb * 64;
    const int Cin = STEM ? 32 : UP ? 64 : p.a.Cs, nchunk = Cin / 32, KST = p.K / 32;      // (STEM: one chunk, known to the compiler - the stage's addresses are not loop invariants to be hoisted and spilled)
    // the map the halo fetch reads: this convolution's input - or, UP, the skip map of the projection that would have produced it (same pixel grid, ps.res's own crop on top)
    const TView& hsrc = UP ? ps.res : p.a;
    const int hsy = (UP ? ps.res.y0 : 0) + p.a.y0 + oy0, hsx = (UP ? ps.res.x0 : 0) + p.a.x0 + ox0;
    const __amdgpu_buffer_rsrc_t A = make_rsrc((const _Float16*)hsrc.p + ((size_t)(b * hsrc.Hs + hsy) * hsrc.Ws + hsx) * Cin, 0x7FFFFFFFu);
    const __amdgpu_buffer_rsrc_t W = make_rsrc((const _Float16*)p.wt_perm + (size_t)nb * KST * 2048, (unsigned)KST * 4096u);   // [KST][4][64][8] of this block
    const unsigned wlane = lane * 16u;

    float4v acc[2][4][4];                                             // [row of the pair][16-pixel group][n-tile]
    auto acc_init = [&](const float* bias) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4v bv = *(const float4v*)(bias + n0 + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1));
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = bv;
        }
    };
    if (!STEM && !UP) acc_init(p.bias);                                            // (STEM: after the halo stage, whose operands need the registers)
    half8 w[3][4];                                                    // ring: slot = ky
    // WLDS (round 6): the chunk's 36 weight fragments are staged in LDS once per workgroup (behind the halo tile) and every wave reads its A operands from there.
    // Until then each wave streamed all 36 KiB of a chunk from L2 through its own register ring - 144 KiB per workgroup and chunk on the vector-memory path beside
    // 42 KiB of halo pixels: with the stream replaced by LDS reads of anything the frame of config 2 took 6.2 instead of 7.6 ms (profiles/r6_kernels/conv3_weights_lds.txt).
    // UP keeps the ring: its staged input pixels take the room (three regions would leave one workgroup per CU).
    constexpr bool WLDS = !UP;
    unsigned char* wsm = smem + C::SMEM;                              // [9 taps][4 n-tiles][64 lanes][16 bytes]
    auto wload = [&](int slot, int kc, int ky, int kx) {     // (ring: past the last chunk the k-step lies beyond the block: the fetch returns zeros, nobody reads them)
        if constexpr (WLDS) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w[slot][nt] = *(const half8*)(wsm + ((ky * 3 + kx) * 4 + nt) * 1024 + wlane);
        } else {
            const unsigned vo = wlane + (unsigned)((ky * 3 + kx) * nchunk + kc) * 4096u;    // k-step (tap * Cin + 32 kc) / 32
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w[slot][nt] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(W, vo + nt * 1024u, 0, 0));
            asm volatile("" ::: "memory");                            // keeps the fetch where it is written
        }
    };
    // staging: thread tid takes 16 bytes of every tap's 4 KiB run (the runs of a chunk lie nchunk x 4 KiB apart in the block's fragment-major copy)
    auto wstage_request = [&](int kc, uint4v (&ws)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t) ws[t] = __builtin_amdgcn_raw_buffer_load_b128(W, (unsigned)tid * 16u, (unsigned)(t * nchunk + kc) * 4096u, 0);
    };
    auto wstage_store = [&](const uint4v (&ws)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t) *(uint4v*)(wsm + t * 4096 + tid * 16) = ws[t];
    };
    auto mm = [&](int r, int slot, const half8 (&xa)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[slot][nt], xa[mt], acc[r][mt][nt], 0, 0, 0);
    };
#ifdef W2X_CONV3_STAGGER   // experiment (tools/ab/lib_variants.sh "k_conv3.hip:-DW2X_CONV3_STAGGER=n"): the workgroups of the launch's first generation that land on the second wave
    if (blockIdx.x < 2u * 256u) {      // slot of their SIMD sleep n x 8128 cycles first, so that the two workgroups of a CU run out of phase (one fetching / storing while the other multiplies)
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID bits 3:0 = wave slot on the SIMD
        if (slot & 1) for (int i = 0; i < W2X_CONV3_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif

    // halo copy: thread (pixel column tid >> 2, 16-byte piece tid & 3) takes its column of every halo row; threads 0..79 also take
    // the two extra columns (64, 65) of row tid >> 3
    const int hrows = min(HR, Ho + 2 - oy0), hcols = min(HC, Wo + 2 - ox0);
    const unsigned rowb = (unsigned)hsrc.Ws * (unsigned)Cin * 2u;
    const int hc0 = tid >> 2, c8 = tid & 3, hr1 = tid >> 3, hc1 = 64 + ((tid >> 2) & 1);
    const unsigned go0 = hc0 < hcols ? (unsigned)(hc0 * Cin + c8 * 8) * 2u : kNoPix;
    const unsigned go1 = (tid < 80 && hr1 < hrows && hc1 < hcols) ? (unsigned)hr1 * rowb + (unsigned)(hc1 * Cin + c8 * 8) * 2u : kNoPix;
    unsigned char* lo0 = smem + halo_slot(hc0, c8);
    unsigned char* lo1 = smem + hr1 * C::ROWB + halo_slot(hc1, c8);
    // fragment reads: lane (fr, g) wants piece g of pixel x = kx + 16 mt + fr of a halo row ((x >> 2) & 3 does not depend on mt)
    const unsigned char* xrow = smem + (2 * wv) * C::ROWB;
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = halo_slot(kx + fr, g);
    // UP: the projection's input pixels under the halo tile, gated, behind the halo tile in LDS ([6][34] pixels, 144 bytes apart)
    constexpr int XR = HR / 2 + 1, XC = HC / 2 + 1, XPIX = XR * XC, XSTRIDE = 144, XIT = (XPIX * 8 + 255) / 256;
    unsigned char* xt = smem + C::SMEM;
    const int Y0 = p.a.y0 + oy0, X0 = p.a.x0 + ox0;                   // the halo tile's origin in the projection's output map
    const int iy0 = Y0 >> 1, ix0 = X0 >> 1;
    uint4v wnext[9];                                                  // WLDS, plain mode: the weights of the chunk to come, in registers until the barrier lets them into LDS
    if constexpr (WLDS && !STEM) wstage_request(0, wnext);
    auto chunk = [&](const int kc) __attribute__((always_inline)) {
        if constexpr (STEM) {
            uint4v wst[9];
            wstage_request(kc, wst);
            asm volatile("" ::: "memory");
            // ---- the chunk (all 32 channels: conv3_stem_supported) computed from the network's input tile: stem_kernel<2>'s arithmetic per group of 16 halo pixels
            constexpr int SNT = 2, NPIX = HR * HC, NGRP = (NPIX + 15) / 16, GPW = (NGRP + 3) / 4, HALF = GPW;             // (HALF: groups of a wave per round)
            const int Hs_o = ps.Mrows / ps.aW, Ws_o = ps.aW;                // extent of the stem's output = this convolution's input map
            const _Float16* __restrict__ Wt = (const _Float16*)ps.wt;
            const half4 zero4h = {};
            half4 wf[SNT][3];
            float4v sbias[SNT];
#pragma unroll
            for (int nt = 0; nt < SNT; ++nt) {
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) {
                    const int tap = 4 * t3 + g;
                    wf[nt][t3] = tap < 9 ? *(const half4*)(Wt + (size_t)(4 * SNT * (fr >> 2) + 4 * nt + (fr & 3)) * ps.Kw + tap * 4) : zero4h;
                }
                sbias[nt] = *(const float4v*)(ps.bias + 4 * SNT * g + 4 * nt);   // accumulator row 4g + j of n-tile nt = channel 8 g + 4 nt + j
            }
            int toff[3];
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int tap = 4 * t3 + g < 9 ? 4 * t3 + g : 8;
                toff[t3] = ((tap / 3) * ps.a.Ws + tap % 3) * 8;      // bytes
            }
            const __amdgpu_buffer_rsrc_t In = make_rsrc((const _Float16*)ps.a.p + (size_t)b * ps.a.Hs * ps.a.Ws * 4, (unsigned)((size_t)ps.a.Hs * ps.a.Ws * 8));   // this tile's input (conv3_stem_supported: < 2 GB)
#pragma unroll
            for (int k0 = 0; k0 < GPW; k0 += HALF) {                        // one round: every operand of the wave's groups is requested before the first product
                half4 xf[HALF][3];
#pragma unroll
                for (int k = 0; k < HALF; ++k) {
                    const int pi = min((wv + 4 * (k0 + k)) * 16 + fr, NPIX - 1), hr = pi / HC, hc = pi - hr * HC;
                    const int Y = min(p.a.y0 + oy0 + hr, Hs_o - 1), X = min(p.a.x0 + ox0 + hc, Ws_o - 1);   // (halo pixels beyond the map feed outputs nobody stores)
                    const unsigned src = (unsigned)(((ps.a.y0 + Y) * ps.a.Ws + ps.a.x0 + X) * 8);
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3) {
                        const half4 v = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(In, src + (unsigned)toff[t3], 0, 0));
                        xf[k][t3] = 4 * t3 + g < 9 ? v : zero4h;
                    }
                }
#pragma unroll
                for (int k = 0; k < HALF; ++k) {
                    const int gi = wv + 4 * (k0 + k), pi = gi * 16 + fr;
                    if (k0 + k >= GPW || gi >= NGRP) break;
                    _Float16 hq[8];
#pragma unroll
                    for (int nt = 0; nt < SNT; ++nt) {
                        float4v a4 = sbias[nt];
#pragma unroll
                        for (int t3 = 0; t3 < 3; ++t3) a4 = __builtin_amdgcn_mfma_f32_16x16x16f16(wf[nt][t3], xf[k][t3], a4, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = a4[j];
                            if (ps.act == 1) v = v > 0.f ? v : v * ps.alpha;
                            hq[4 * nt + j] = (_Float16)v;
                        }
                    }
                    const int hr = pi / HC, hc = pi - hr * HC;
                    if (pi < NPIX) *(half8*)(smem + hr * C::ROWB + halo_slot(hc, g)) = (half8){hq[0], hq[1], hq[2], hq[3], hq[4], hq[5], hq[6], hq[7]};
                }
            }
            wstage_store(wst);
            const float* bias_late = p.bias;
            asm volatile("" : "+s"(bias_late) :: "memory");                 // the accumulators are set up HERE (hoisted above the stage they cost it 39 spilled registers)
            acc_init(bias_late);
        } else if constexpr (!UP) {
            uint4v h[HR], h1;
#pragma unroll
            for (int hr = 0; hr < HR; ++hr)
                h[hr] = __builtin_amdgcn_raw_buffer_load_b128(A, hr < hrows ? go0 : kNoPix, (unsigned)hr * rowb + (unsigned)kc * 64u, 0);
            h1 = __builtin_amdgcn_raw_buffer_load_b128(A, go1, (unsigned)kc * 64u, 0);
            __syncthreads();                                   // the previous chunk's products are done with the halo tile and the weights
#pragma unroll
            for (int hr = 0; hr < HR; ++hr) *(uint4v*)(lo0 + hr * C::ROWB) = h[hr];
            if (tid < 80) *(uint4v*)lo1 = h1;
            wstage_store(wnext);                               // (requested under the previous chunk's products; chunk 0: in front of the loop)
        }
        if constexpr (UP) {
            // ---- the chunk's halo tile = channels 32 kc .. + 31 of the projection + the skip map, assembled by sub-pixel class: wave w owns class (w >> 1, w & 1), 165 halo
            // pixels = eleven groups of 16; lane (fr, g) ends with piece g of pixel fr of each group.  The lane's skip pieces are requested first and arrive under the products.
            // Groups: class row r5 (halo row 2 r5 + r_off) holds 33 pixels = two groups of 16 (k = 2 r5 + j: class columns 16 j + fr) and one left over; the five left-over
            // pixels (class column 32) are group 10, pixel fr = class row fr.  Every address is then a per-lane constant plus a compile-time / wave-uniform term.
            constexpr int NGR = 11;
            const int py = wv >> 1, px = wv & 1, sub = wv;
            const int r_off = (py - Y0) & 1, c_off = (px - X0) & 1;        // first halo row / column of the class
            int frl = fr;
            asm volatile("" : "+v"(frl));                                   // (per-chunk addresses: as loop invariants they would be hoisted and spilled)
            const int hc_l = 2 * frl + c_off;                               // halo column of pixel fr in group j = 0 (j = 1: + 32; left-over group: 64 + c_off)
            const int hr_x = 2 * min(frl, HR / 2 - 1) + r_off;              // halo row of pixel fr in the left-over group
            // weights [4 x 64][Kw] (row (dy * 2 + dx) * 64 + channel) and bias through buffer resources: as pointer loads the compiler sank them below the barrier
            const __amdgpu_buffer_rsrc_t WQ = make_rsrc(ps.wt, 256u * (unsigned)ps.Kw * 2u), BQ = make_rsrc(ps.bias, 256u * 4u);
            half8 wq[2][2];
            float4v bq[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {                                // A fragment row fr = 4 g' + j of n-tile nt = channel 8 g' + 4 nt + j of the chunk
                const int n = sub * 64 + kc * 32 + 8 * (frl >> 2) + 4 * nt + (frl & 3);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wq[nt][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(WQ, (unsigned)(n * ps.Kw + ks * 32 + g * 8) * 2u, 0, 0));
                bq[nt] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(BQ, (unsigned)(sub * 64 + kc * 32 + 8 * g + 4 * nt) * 4u, 0, 0));
            }
            asm volatile("" ::: "memory");                                  // weights first: memory returns in order, the products wait for these only
            const float slope = ps.act == 1 ? ps.alpha : 1.f;               // LeakyReLU with a slope in [0, 1] (conv3_up_supported) as max(v, v * slope) = pixgemm_kernel's v > 0 ? v : v * alpha; 1: none
            // skip pieces: piece g of the lane's pixel in every group (rows / columns beyond the map: zeros)
            const unsigned so_l = hc_l < hcols ? (unsigned)(hc_l * Cin + g * 8) * 2u : kNoPix;                  // groups j = 0; j = 1 adds 32 pixels
            const unsigned so_h = hc_l + 32 < hcols ? so_l + 32u * (unsigned)Cin * 2u : kNoPix;
            const unsigned so_x = (frl < HR / 2 && hr_x < hrows && 64 + c_off < hcols) ? (unsigned)hr_x * rowb + (unsigned)((64 + c_off) * Cin + g * 8) * 2u : kNoPix;
            uint4v sk[NGR];
#pragma unroll
            for (int k = 0; k < NGR - 1; ++k) {
                const int hr = 2 * (k >> 1) + r_off;
                sk[k] = __builtin_amdgcn_raw_buffer_load_b128(A, hr < hrows ? ((k & 1) ? so_h : so_l) : kNoPix, (unsigned)hr * rowb + (unsigned)kc * 64u, 0);
            }
            sk[NGR - 1] = __builtin_amdgcn_raw_buffer_load_b128(A, so_x, (unsigned)kc * 64u, 0);
            asm volatile("" ::: "memory");
            if (kc == 0) {   // the projection's input pixels: requested behind the first chunk's skip pieces (one round trip for both), gated, to LDS behind the halo tile
                const int Hl = ps.Mrows / ps.aW, Wl = ps.aW;
                const __amdgpu_buffer_rsrc_t XB = make_rsrc((const _Float16*)ps.a.p + (size_t)b * ps.Mrows * 64, (unsigned)ps.Mrows * 128u);
                float sc[8];                                                    // the gate of this thread's pieces: all of them hold channels 8 (tid & 7) .. + 7
                if (ps.a_scale) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) sc[j] = ps.a_scale[(size_t)b * 64 + (tid & 7) * 8 + j];
                }
                uint4v xr[XIT];
#pragma unroll
                for (int k = 0; k < XIT; ++k) {
                    const int idx = k * 256 + tid, pix = idx >> 3, c = idx & 7, ly = pix / XC, lx = pix - ly * XC;
                    const bool in = idx < XPIX * 8 && iy0 + ly < Hl && ix0 + lx < Wl;
                    xr[k] = __builtin_amdgcn_raw_buffer_load_b128(XB, in ? (unsigned)(((iy0 + ly) * Wl + ix0 + lx) * 128 + c * 16) : kNoPix, 0, 0);
                }
                if (ps.a_scale) {                                              // fp16(x * s), the rounding of pixgemm_kernel's rows
#pragma unroll
                    for (int k = 0; k < XIT; ++k) xr[k] = __builtin_bit_cast(uint4v, gate::gate8(__builtin_bit_cast(half8, xr[k]), sc));
                }
#pragma unroll
                for (int k = 0; k < XIT; ++k) {
                    const int idx = k * 256 + tid, pix = idx >> 3, c = idx & 7;
                    if (idx < XPIX * 8) *(uint4v*)(xt + pix * XSTRIDE + c * 16) = xr[k];
                }
                asm volatile("" ::: "memory");
            }
            __syncthreads();                                                // the previous chunk's products are done with the halo tile (first chunk: the staged input pixels are in LDS)
            // LDS: the lane's pixel of a group in the staged input tile (B operand, 2 x 16 bytes) and in the halo tile (one 16-byte store)
            const int xl = (((X0 + hc_l) >> 1) - ix0) * XSTRIDE + g * 16;                                     // j = 1: + 16 pixels
            const int xx = ((((Y0 + hr_x) >> 1) - iy0) * XC + ((X0 + 64 + c_off) >> 1) - ix0) * XSTRIDE + g * 16;
            const int hl = halo_slot(hc_l, g);                                                                // j = 1: + 32 * 64 bytes (the rotation has period 16 pixels)
            const int hx = hr_x * C::ROWB + halo_slot(64 + c_off, g);
            if (kc == 0) {
                const float* bias_late = p.bias;
                asm volatile("" : "+s"(bias_late) :: "memory");             // (as in the STEM stage: the accumulators are set up behind the first chunk's requests)
                acc_init(bias_late);
            }
#pragma unroll
            for (int k = 0; k < NGR; ++k) {
                const bool last = k == NGR - 1;
                const int hr = 2 * (k >> 1) + r_off;                                                         // (groups 0 .. 9)
                const unsigned char* xp = xt + (last ? xx : (((Y0 + hr) >> 1) - iy0) * (XC * XSTRIDE) + xl + (k & 1) * 16 * XSTRIDE);
                unsigned char* hp = smem + (last ? hx : hr * C::ROWB + hl + (k & 1) * 32 * 64);
                if (k == 8) { wload(0, kc, 0, 0); wload(1, kc, 1, 0); }     // the first taps of this chunk's products, into registers the skip pieces have left
                const half8 xb0 = *(const half8*)xp, xb1 = *(const half8*)(xp + 64);
                float4v a4[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    a4[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[nt][0], xb0, bq[nt], 0, 0, 0);
                    a4[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[nt][1], xb1, a4[nt], 0, 0, 0);
                    a4[nt] = __builtin_elementwise_max(a4[nt], a4[nt] * slope);
                }
                half8 o = (half8){(_Float16)a4[0][0], (_Float16)a4[0][1], (_Float16)a4[0][2], (_Float16)a4[0][3], (_Float16)a4[1][0], (_Float16)a4[1][1], (_Float16)a4[1][2], (_Float16)a4[1][3]};
                o += __builtin_bit_cast(half8, sk[k]);                      // fp16 + fp16, as pixgemm_kernel's epilogue
                if (!last || frl < HR / 2) *(half8*)hp = o;
            }
        }
        __syncthreads();
        if constexpr (WLDS) { wload(0, kc, 0, 0); wload(1, kc, 1, 0); }
        if constexpr (WLDS && !STEM) {                                  // the next chunk's weights travel under this chunk's products
            if (kc + 1 < nchunk) wstage_request(kc + 1, wnext);         // (uniform)
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            half8 xa[4];
            auto xload = [&](int hh) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(xrow + xoff[kx] + hh * C::ROWB + mt * 1024);
            };
            xload(0);
            mm(0, 0, xa);
            xload(1);
            wload(2, kc, 2, kx);
            mm(0, 1, xa); mm(1, 0, xa);
            xload(2);
            if (kx < 2) wload(0, kc, 0, kx + 1);
            mm(0, 2, xa); mm(1, 1, xa);
            xload(3);
            if (kx < 2) wload(1, kc, 1, kx + 1);
            mm(1, 2, xa);
        }
    };
    if constexpr (UP) { chunk(0); chunk(1); }                         // (two chunks, each with its own register allocation: the first one also stages the projection's input pixels)
    else {
#pragma unroll 1
        for (int kc = 0; kc < nchunk; ++kc) chunk(kc);
    }

    // ---- epilogue: LeakyReLU / none, fp16, 2 x 16 bytes per lane and pixel; squeeze-excite partial sums of the stored values
    const __amdgpu_buffer_rsrc_t O = make_rsrc((_Float16*)p.out.p + (size_t)b * p.out.Hs * p.out.Ws * p.out.Cs, 0x7FFFFFFFu);
    const float slope = p.act == 1 ? p.alpha : 1.f;                    // LeakyReLU as max(v, v * alpha), 0 <= alpha <= 1 (conv3_supported); 1: none
    float csum[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) csum[nt][j] = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + 2 * wv + r;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int ox = ox0 + mt * 16 + fr;
            const bool valid = oy < Ho && ox < Wo;
            _Float16 hv[16];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[r][mt][nt][j];
                    hv[nt * 4 + j] = (_Float16)fmaxf(v, v * slope);
                    if (POOL && valid) csum[nt][j] += (float)hv[nt * 4 + j];
                }
            const unsigned oo = valid ? ((unsigned)(oy * p.out.Ws + ox) * (unsigned)p.out.Cs + (unsigned)(n0 + 8 * g)) * 2u : kNoPix;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, (half8){hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]}), O, oo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, (half8){hv[8], hv[9], hv[10], hv[11], hv[12], hv[13], hv[14], hv[15]}), O,
                                                   __builtin_elementwise_add_sat(oo, 64u), 0, 0);
        }
    }
    if (POOL) {   // per-workgroup partial sums in a fixed order (pixels of a lane, lanes of a row group, waves 0..3); se_kernel adds the tiles of an image
        __syncthreads();                                        // all waves are done with the halo tile
        float* ws = (float*)smem;                               // [4][64]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = csum[nt][j];
                t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
                if (fr == 0) ws[wv * 64 + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1) + j] = t;
            }
        __syncthreads();
        if (tid < 64) p.pool_out[(size_t)tidx * p.out.Cs + n0 + tid] = ws[tid] + ws[64 + tid] + ws[128 + tid] + ws[192 + tid];
    }
}

}  // namespace

int conv3_tiles(const GemmParams& p) {   // workgroups (pooling partials) per image and 64-channel block
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    return ((Wo + Conv3Cfg::TW - 1) / Conv3Cfg::TW) * ((Ho + Conv3Cfg::TH - 1) / Conv3Cfg::TH);
}

bool conv3_supported(const GemmParams& p) {
    if (switches().no_conv3 || p.a_scale || p.res_scale || !p.wt_perm || p.amode != 2 || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.omode != 0 || p.ln || (p.act != 0 && p.act != 1) ||
        p.has_clip || p.stats_out || p.res.p || p.res2.p) return false;
    if (p.act == 1 && !(p.alpha >= 0.f && p.alpha <= 1.f)) return false;   // LeakyReLU as max(v, v * alpha)
    const int Cin = p.a.Cs;
    if (p.K != 9 * Cin || p.Kw != p.K || Cin % 32 || p.out.Cs != p.N || p.aW <= 0 || p.Mrows % p.aW) return false;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    if (p.a.y0 + Ho + 2 > p.a.Hs || p.a.x0 + Wo + 2 > p.a.Ws || p.out.Hs < Ho || p.out.Ws < Wo) return false;
    if ((size_t)p.out.Hs * p.out.Ws * p.out.Cs * 2 > 0x7FFFFFFFull || (size_t)(Conv3Cfg::HR + 1) * p.a.Ws * Cin * 2 > 0x7FFFFFFFull) return false;   // 32-bit offsets
    // pooling partials: one per workgroup; the plan sized the buffer for ceil(Mrows / kGemmBM) row tiles per image
    if (p.pool_out && conv3_tiles(p) > (p.Mrows + kGemmBM - 1) / kGemmBM) return false;
    return p.N == 64 || p.N == 128 || p.N == 256;
}

namespace {
constexpr int kWldsSmem = Conv3Cfg::SMEM + Conv3Cfg::WCHUNK;                                  // halo tile + a chunk's weights (conv3_kernel, WLDS): 79 104 bytes, two workgroups per CU
constexpr int kUpSmem = Conv3Cfg::SMEM + (Conv3Cfg::HR / 2 + 1) * (Conv3Cfg::HC / 2 + 1) * 144;   // halo tile + the projection's staged input pixels (conv3_kernel, UP)
hipError_t launch_conv3_any(const GemmParams& p, const GemmParams* ps, int mode, hipStream_t s) {
    using C = Conv3Cfg;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    static unsigned lds_ok = 0, lds_ok_pool = 0, lds_ok_stem = 0, lds_ok_up = 0;   // per-device bits: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3_kernel<false, 0>, kWldsSmem, lds_ok); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3_kernel<true, 0>, kWldsSmem, lds_ok_pool); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3_kernel<false, 1>, kWldsSmem, lds_ok_stem); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3_kernel<false, 2>, kUpSmem, lds_ok_up); e != hipSuccess) return e;
    const int tiles_x = (Wo + C::TW - 1) / C::TW, tiles_y = (Ho + C::TH - 1) / C::TH, nblk = p.N / 64;
    const int xcd_order = nblk > 1 ? 1 : 0;
    const int ntiles = p.B * tiles_x * tiles_y;
    const dim3 grid((unsigned)((xcd_order ? (ntiles + 7) / 8 * 8 : ntiles) * nblk));
    if (mode == 1) hipLaunchKernelGGL((conv3_kernel<false, 1>), grid, dim3(256), kWldsSmem, s, p, Ho, Wo, tiles_x, tiles_y, nblk, xcd_order, *ps);
    else if (mode == 2) hipLaunchKernelGGL((conv3_kernel<false, 2>), grid, dim3(256), kUpSmem, s, p, Ho, Wo, tiles_x, tiles_y, nblk, xcd_order, *ps);
    else if (p.pool_out) hipLaunchKernelGGL((conv3_kernel<true, 0>), grid, dim3(256), kWldsSmem, s, p, Ho, Wo, tiles_x, tiles_y, nblk, xcd_order, GemmParams{});
    else hipLaunchKernelGGL((conv3_kernel<false, 0>), grid, dim3(256), kWldsSmem, s, p, Ho, Wo, tiles_x, tiles_y, nblk, xcd_order, GemmParams{});
    return hipGetLastError();
}
}  // namespace

hipError_t launch_conv3(const GemmParams& p, hipStream_t s) { return launch_conv3_any(p, nullptr, 0, s); }

// the stem (ps: a launch stem_supported() takes, 32 output channels) folded into the 32 -> 64 convolution that is the only reader of its output (p.a = a view of ps.out)
bool conv3_stem_supported(const GemmParams& p, const GemmParams& ps) {
    if (switches().no_fuse_stem || !conv3_supported(p) || !stem_supported(ps) || p.pool_out || p.a.Cs != 32 || p.N != 64 || ps.N != 32 || ps.out.Cs != 32 || p.a.p != ps.out.p || ps.out.y0 || ps.out.x0) return false;
    const int Hs_o = ps.Mrows / ps.aW, Ws_o = ps.aW, Ho = p.Mrows / p.aW, Wo = p.aW;
    return p.a.Hs == ps.out.Hs && p.a.Ws == ps.out.Ws && p.a.y0 >= 0 && p.a.x0 >= 0 && p.a.y0 + Ho + 2 <= Hs_o && p.a.x0 + Wo + 2 <= Ws_o && p.B == ps.B;
}

hipError_t launch_conv3_stem(const GemmParams& p, const GemmParams& ps, hipStream_t s) { return launch_conv3_any(p, &ps, 1, s); }

// the pixel-shuffle projection q (cunet's ConvTranspose 2x2 stride 2 as Linear 64 -> 4 x 64 + DepthToSpace, LeakyReLU, gate on its rows, skip add: a launch
// pixgemm_supported() takes) folded into the 64 -> 64 convolution p that is the only reader of its output (p.a = a view of q.out)
bool conv3_up_supported(const GemmParams& p, const GemmParams& q) {
    if (switches().no_fuse_up || !conv3_supported(p) || !pixgemm_supported(q) || p.pool_out || p.a.Cs != 64 || p.N != 64) return false;
    if (q.act == 1 && !(q.alpha >= 0.f && q.alpha <= 1.f)) return false;   // LeakyReLU as max(v, v * alpha)
    if (q.omode != 2 || q.r != 2 || q.K != 64 || q.N != 256 || q.out.Cs != 64 || !q.wt || q.Kw < 64 || !q.res.p || q.res.Cs != 64 || q.res_scale || q.res2.p || q.a.Cs != 64) return false;
    if (p.a.p != q.out.p || q.out.y0 || q.out.x0 || p.a.Hs != q.out.Hs || p.a.Ws != q.out.Ws || p.B != q.B || q.aW <= 0 || q.Mrows % q.aW) return false;
    const int Hl = q.Mrows / q.aW, Wl = q.aW, Ho = p.Mrows / p.aW, Wo = p.aW;
    if (p.a.y0 < 0 || p.a.x0 < 0 || p.a.y0 + Ho + 2 > 2 * Hl || p.a.x0 + Wo + 2 > 2 * Wl || q.out.Hs < 2 * Hl || q.out.Ws < 2 * Wl) return false;   // every stored output reads pixels the projection produces
    if (q.res.y0 < 0 || q.res.x0 < 0 || q.res.y0 + p.a.y0 + Ho + 2 > q.res.Hs || q.res.x0 + p.a.x0 + Wo + 2 > q.res.Ws) return false;
    return (size_t)(Conv3Cfg::HR + 1) * q.res.Ws * 64 * 2 <= 0x7FFFFFFFull && (size_t)q.Mrows * 128 < 0xFFFF0000ull;      // 32-bit offsets
}

hipError_t launch_conv3_up(const GemmParams& p, const GemmParams& q, hipStream_t s) { return launch_conv3_any(p, &q, 2, s); }

}  // namespace w2x
