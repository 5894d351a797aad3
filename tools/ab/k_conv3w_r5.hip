// Direct 3x3 convolution (stride 1, valid) for gfx950, input maps of 64 / 128 / 256 channels (cunet) - the WIDE-CHUNK, persistent form of k_conv3.hip:
//     out[b][y][x][n] = act( sum_{ky,kx,c} in[b][y+ky][x+kx][c] * W[n][(ky*3+kx)*Cin + c] + bias[n] )
//
// Why a second kernel.  k_conv3.hip stages the halo tile of ONE 32-channel chunk at a time: of a 64-channel pixel (128 bytes = one cache line) it
// requests 64 bytes now and the other 64 a few microseconds later.  tools/ab/halfline_bench.hip, same tiles, same bytes: two 64-byte halves a chunk
// apart move 4.1 TB/s, whole 128-byte pixels 7.1 TB/s - and round 2's ablations had the memory side of the 64 -> 64 layer alone at 0.74 of its 1.03 ms.
// Here a chunk is 64 channels, so every request is a whole line:
//   * one persistent workgroup of EIGHT waves per CU walks over (tile, 64-channel block) items; tile = 8 rows x 60 columns of one block, halo image
//     10 rows x 62 pixels x 128 bytes = 77.5 KB, TWO of them (2 x 79 360 B + 2 KB of pooling partials: 160 KB of LDS to the last kilobyte): the next
//     chunk's image - or the next item's first - is in flight by LDS-DMA (buffer_load ... lds: no registers, no ds_write; eight lanes fetch the 128
//     contiguous bytes of a pixel) while the products of the current one run, the stores of a finished tile are issued under the next products;
//   * waves split the tile by rows AND by output channels: wave v owns rows 2 (v & 3), 2 (v & 3) + 1 and the n-tiles 2 (v >> 2), 2 (v >> 2) + 1 of the
//     block (64 accumulator registers).  A weight fragment still serves 2 rows x 4 pixel groups = the reuse of k_conv3.hip, so the weight stream from
//     L2 (fragorder.h frag_conv3b, a ring of three taps) is the same bytes per product; a halo fragment is read by the two waves that share its rows;
//   * halo image in LDS: pixel P = row * 62 + x at byte 128 P, its eight 16-byte pieces rotated by 2 ((x >> 1) & 3) slots.  A ds_read_b128 is served in
//     groups of 16 lanes that pair the k-groups (g = 0 with 1, 2 with 3) over complementary pixel sets ({0-3, 12-15} with {4-11}) on 64 banks = two
//     pixels; of the eight pixels of a group four are even and four odd, and the rotation sends the four of one parity that read piece g and the four
//     that read piece g + 1 to eight different slots: conflict-free.  The DMA writes lane-contiguously and fetches, per lane, the piece that belongs
//     in its slot;
//   * everything else as in k_conv3.hip: transposed product with permuted A rows (a lane ends with 8 consecutive channels of a pixel: one 16-byte
//     store), bias as the initial accumulator, LeakyReLU as max(v, v alpha), squeeze-excite pooling partials per tile in a fixed order.
// 60 valid columns of the 64 computed: cunet's extents (442, 218, 106 ...) take the same number of tiles as with 62 or 64.
// Results differ from k_conv3.hip's only in the order of the fp32 sums over channels (two 32-channel steps per tap instead of a chunk per pass).
#include "kernels.h"

#include <algorithm>
#include <cstring>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr unsigned kNoPix = 0xFFFFFFFFu;     // buffer offset past every resource: reads zeros, drops stores
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);   // raw buffer, 32-bit offsets, bounds-checked
}

struct Conv3wCfg {
    static constexpr int TH = 8, TWO = 60, HR = TH + 2, HP = TWO + 2, NWV = 8, KC = 64;
    static constexpr int PIXB = KC * 2;                                 // bytes per halo pixel
    static constexpr int BUF = HR * HP * PIXB;                          // 79 360
    static constexpr int NDMA = (BUF + 1023) / 1024;                    // wave instructions per image (the last one half empty)
    static constexpr int POOLB = 3 * 4 * 64 * 4;                        // pooling partials [3 tiles in flight][4 row pairs][64 channels] floats
    static constexpr int SMEM = 2 * BUF + POOLB;                        // 161 792 <= 163 840
};
struct C3wItem { int b, oy0, ox0, nb, tidx; };

// Diagnostic build only (-DW2X_C3W_STAMPS): s_memtime deltas per phase, one record per wave, plain stores.  0 wait + barrier at the top of a chunk, 1 DMA issue,
// 2 epilogue, 3 bias + products, 4 chunks, 5 waves
#ifdef W2X_C3W_STAMPS
__device__ unsigned long long g_c3w_stamps[256 * 8][8];
#define W2X_STAMP(K) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); tacc[K] += t_ - tprev; tprev = t_; }
#else
#define W2X_STAMP(K)
#endif

template <bool POOL>
__global__ __launch_bounds__(512, 1) void conv3w_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, int nblk, int ntiles) {
    using C = Conv3wCfg;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wv & 3, nh = wv >> 2;                                // row pair, half of the block's output channels
    const int fr = lane & 15, g = lane >> 4;
    const int Cin = p.a.Cs, nchunk = Cin / C::KC, KST = p.K / 32, kpt = Cin / 32;   // k-steps per tap
    const int tpi = tiles_x * tiles_y, G = gridDim.x, nitems = ntiles * nblk;
    // this workgroup's items: all blocks of tile blockIdx.x, then of tile blockIdx.x + G, ... (a tile's blocks re-read its halo through this XCD's L2)
    auto decode = [&](int c) {
        C3wItem t;
        t.nb = c % nblk;
        t.tidx = (int)blockIdx.x + (c / nblk) * G;
        t.b = t.tidx / tpi;
        const int trem = t.tidx - t.b * tpi, ty = trem / tiles_x, tx = trem - ty * tiles_x;
        t.oy0 = ty * C::TH; t.ox0 = tx * C::TWO;
        return t;
    };
    const int my_items = blockIdx.x < (unsigned)ntiles ? ((ntiles - 1 - (int)blockIdx.x) / G + 1) * nblk : 0;
    const size_t img_a = (size_t)p.a.Hs * p.a.Ws * Cin * 2, all_a = img_a * p.B;
    const unsigned rowb = (unsigned)p.a.Ws * (unsigned)Cin * 2u;
    // halo fetch of one chunk: instruction q (of NDMA; wave v issues q = v, v + 8, ...: ten each) fills LDS bytes [1024 q, 1024 q + 1024) = pixels 8 q .. 8 q + 7; lane =
    // (pixel lane >> 3, slot lane & 7) fetches piece (slot - rot(x)) & 7 of pixel P = 8 q + (lane >> 3) = (row P / 62, x P % 62).  The lane's part of the address does
    // not depend on the tile: worked out once.
    constexpr int NQ = (C::NDMA + C::NWV - 1) / C::NWV;
    unsigned dma_vo[NQ]; int dma_row[NQ], dma_x[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = wv + C::NWV * i;
        const int P = 8 * q + (lane >> 3), row = P / C::HP, x = P - row * C::HP;
        const int piece = ((lane & 7) - 2 * ((x >> 1) & 3)) & 7;
        dma_row[i] = (q < C::NDMA && P < C::HR * C::HP) ? row : 1 << 20;      // (past the image: never fetched - the lanes stay off, they would write into the other buffer)
        dma_x[i] = x;
        dma_vo[i] = (unsigned)row * rowb + (unsigned)(x * Cin * 2 + piece * 16);
    }
    auto dma = [&](const C3wItem& t, int kc, int buf) {
        const size_t org = (size_t)t.b * img_a + ((size_t)(p.a.y0 + t.oy0) * p.a.Ws + p.a.x0 + t.ox0) * Cin * 2;
        const size_t left = all_a - org;
        const __amdgpu_buffer_rsrc_t A = make_rsrc((const unsigned char*)p.a.p + org, left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)left);
        const int hrows = min(C::HR, Ho + 2 - t.oy0), hcols = min(C::HP, Wo + 2 - t.ox0);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = wv + C::NWV * i;
            const unsigned vo = (dma_row[i] < hrows && dma_x[i] < hcols) ? dma_vo[i] : kNoPix;      // outside the map: zeros
            if (dma_row[i] < C::HR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(A, (__attribute__((address_space(3))) void*)(smem + buf * C::BUF + q * 1024), 16, vo, (unsigned)(kc * C::PIXB), 0, 0);
        }
    };
    const __amdgpu_buffer_rsrc_t W = make_rsrc(p.wt_perm, (unsigned)p.N * (unsigned)p.K * 2u);   // [N/64][KST][4][64][8]
    const unsigned wlane = lane * 16u + (unsigned)(2 * nh) * 1024u;    // this wave's two n-tiles of a k-step
    half8 w[3][2];                                                      // ring: slot = ky; [n-tile of the pair]
    auto wload = [&](int slot, int nb, int kc, int ky, int kx, int ks, bool any) {   // !any: offset past the matrix, the fetch returns zeros nobody reads
        const unsigned vo = any ? wlane + (unsigned)(nb * KST + (ky * 3 + kx) * kpt + 2 * kc + ks) * 4096u : kNoPix;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w[slot][nt] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(W, __builtin_elementwise_add_sat(vo, nt * 1024u), 0, 0));
        asm volatile("" ::: "memory");                                  // keeps the fetch where it is written
    };
    float4v acc[2][4][2];                                               // [row of the pair][16-pixel group][n-tile of the pair]
    auto mm = [&](int r, int slot, const half8 (&xa)[4]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[slot][nt], xa[mt], acc[r][mt][nt], 0, 0, 0);
    };
    const float slope = p.act == 1 ? p.alpha : 1.f;                    // LeakyReLU as max(v, v * alpha), 0 <= alpha <= 1 (conv3w_supported); 1: none
    const size_t img_o = (size_t)p.out.Hs * p.out.Ws * p.out.Cs * 2;
    auto epilogue = [&](const C3wItem& t, int seq) {   // fp16, 16 bytes per lane and pixel (this wave's 32 channels: 8 per lane group); squeeze-excite partial sums of the stored values
        const __amdgpu_buffer_rsrc_t O = make_rsrc((unsigned char*)p.out.p + (size_t)t.b * img_o, (unsigned)img_o);
        const int n0 = t.nb * 64 + 32 * nh, xend = min(Wo, t.ox0 + C::TWO);
        float csum[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[nt][j] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = t.oy0 + 2 * rp + r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ox = t.ox0 + mt * 16 + fr;
                const bool valid = oy < Ho && ox < xend;
                _Float16 hv[8];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = acc[r][mt][nt][j];
                        hv[nt * 4 + j] = (_Float16)fmaxf(v, v * slope);
                        if (POOL && valid) csum[nt][j] += (float)hv[nt * 4 + j];
                    }
                const unsigned oo = valid ? ((unsigned)(oy * p.out.Ws + ox) * (unsigned)p.out.Cs + (unsigned)(n0 + 8 * g)) * 2u : kNoPix;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, (half8){hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]}), O, oo, 0, 0);
            }
        }
        if (POOL) {   // per-tile partial sums in a fixed order (pixels of a lane, lanes of a row group, then row pairs 0..3 in pool_reduce)
            float* ws = (float*)(smem + 2 * C::BUF) + (seq % 3) * 256;        // [4 row pairs][64 channels of the block], a ring of three tiles
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sum = csum[nt][j];
                    sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
                    if (fr == 0) ws[rp * 64 + 32 * nh + 8 * g + 4 * nt + j] = sum;
                }
        }
    };
    // the partials of item `seq` are complete once both halves of the workgroup have run its epilogue and a barrier has passed; wave 4 adds the row pairs
    auto pool_reduce = [&](const C3wItem& t, int seq) {
        if (POOL && wv == 4) {
            const float* ws = (const float*)(smem + 2 * C::BUF) + (seq % 3) * 256;
            p.pool_out[(size_t)t.tidx * p.out.Cs + t.nb * 64 + lane] = ws[lane] + ws[64 + lane] + ws[128 + lane] + ws[192 + lane];
        }
    };

    if (my_items == 0) return;
    // Two roles, so that one wave of a SIMD multiplies while the other moves data (waves v and v + 4 share a SIMD; a period = one chunk, between two barriers):
    //   waves 0..3 (A):  their half of the next chunk's DMA  ->  products  ->  the tile's epilogue when this was its last chunk
    //   waves 4..7 (B):  the PREVIOUS tile's epilogue (its accumulators wait for it)  ->  their half of the DMA  ->  products
    // Measured with everybody doing everything in the same order (stamps, tools/ab/c3w_stamps.py): 20 K cycles per chunk for 9.2 K of products.
#ifdef W2X_C3W_STAMPS
    unsigned long long tprev, tacc[5] = {};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
    int c = 0, kc = 0, buf = 0, seq = 0;                 // seq: items this wave has finished (= the slot of the pooling ring it writes next)
    C3wItem cur = decode(0), prev = cur, red = cur;
    bool pend = false; int red_seq = -1;
    dma(cur, 0, 0);
    wload(0, cur.nb, 0, 0, 0, 0, true);
    wload(1, cur.nb, 0, 1, 0, 0, true);
    // fragment reads: lane (fr, g) wants piece 4 ks + g of pixel x = kx + 16 mt + fr of a halo row; the last columns of the computed 64 (outputs 60 .. 63: never
    // stored) would read pixels past the 62-pixel row: they read pixel 61 instead
    int xoff[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int x = min(kx + 16 * mt + fr, C::HP - 1);
            xoff[kx][mt] = x * C::PIXB + ((g + 2 * ((x >> 1) & 3)) & 7) * 16;       // + 64 bytes (pieces 4 .. 7) for ks = 1, modulo the pixel: below
        }
    const unsigned char* xlane = smem + (2 * rp) * C::HP * C::PIXB;
    for (;;) {
        const bool live = c < my_items;
        int nc = c, nkc = kc + 1;
        if (nkc == nchunk) { nkc = 0; nc = c + 1; }
        const bool more = nc < my_items;
        const C3wItem nxt = nkc == 0 ? (more ? decode(nc) : cur) : cur;
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's part of the halo image has landed (and the ring's first taps, and earlier stores)
        __syncthreads();                                   // ... everybody's; everybody is done with the other buffer (and with the pooling partials of two tiles ago)
        W2X_STAMP(0)
        if (red_seq >= 0) { pool_reduce(red, red_seq); red_seq = -1; }
        if (nh == 0) { if (live && more) dma(nxt, nkc, buf ^ 1); W2X_STAMP(1) }
        else {
            if (kc == 0 && pend) { epilogue(prev, seq); red = prev; red_seq = seq; ++seq; pend = false; }
            W2X_STAMP(2)
            if (live && more) dma(nxt, nkc, buf ^ 1);
            W2X_STAMP(1)
        }
        if (!live) break;
        if (kc == 0) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float4v bv = *(const float4v*)(p.bias + cur.nb * 64 + 32 * nh + 8 * g + 4 * nt);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = bv;
            }
        }
        const unsigned char* xb = xlane + buf * C::BUF;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 xa[4];
                auto xload = [&](int hh) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int o = xoff[kx][mt];
                        xa[mt] = *(const half8*)(xb + hh * C::HP * C::PIXB + (ks ? (o & ~127) | ((o + 64) & 127) : o));
                    }
                };
                const bool last = kx == 2 && ks == 1;
                xload(0);
                mm(0, 0, xa);
                xload(1);
                wload(2, cur.nb, kc, 2, kx, ks, true);
                mm(0, 1, xa); mm(1, 0, xa);
                xload(2);
                if (!last) wload(0, cur.nb, kc, 0, ks ? kx + 1 : kx, ks ^ 1, true); else wload(0, nxt.nb, nkc, 0, 0, 0, more);
                mm(0, 2, xa); mm(1, 1, xa);
                xload(3);
                if (!last) wload(1, cur.nb, kc, 1, ks ? kx + 1 : kx, ks ^ 1, true); else wload(1, nxt.nb, nkc, 1, 0, 0, more);
                mm(1, 2, xa);
            }
        }
        W2X_STAMP(3)
#ifdef W2X_C3W_STAMPS
        tacc[4] += 1;
#endif
        if (kc == nchunk - 1) {
            if (nh == 0) { epilogue(cur, seq); ++seq; W2X_STAMP(2) }      // (A: now, under B's products; the reduction of this tile waits for B's half: red / red_seq are B's, set when B gets there)
            else { prev = cur; pend = true; }
        }
        c = nc; kc = nkc; cur = nxt; buf ^= 1;
    }
#ifdef W2X_C3W_STAMPS
    if (lane == 0 && blockIdx.x < 256) { for (int k = 0; k < 5; ++k) g_c3w_stamps[blockIdx.x * 8 + wv][k] += tacc[k]; g_c3w_stamps[blockIdx.x * 8 + wv][5] += 1; }
#endif
    if (POOL) {      // the last tile's partials: B's half was written in the closing period
        __syncthreads();
        if (red_seq >= 0) pool_reduce(red, red_seq);
    }
}

}  // namespace

#ifdef W2X_C3W_STAMPS
// out[2][6]: phase sums over waves 0..3 / waves 4..7 (the DMA issuers), then cleared
extern "C" void w2x_c3w_stamps(unsigned long long* out) {
    static unsigned long long h[256 * 8][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3w_stamps), sizeof(h));
    for (int k = 0; k < 12; ++k) out[k] = 0;
    for (int w = 0; w < 256 * 8; ++w) for (int k = 0; k < 6; ++k) out[((w & 7) >= 4 ? 6 : 0) + k] += h[w][k];
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c3w_stamps), h, sizeof(h));
}
#endif

int conv3w_tiles(const GemmParams& p) {   // pooling partials per image and 64-channel block
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    return ((Wo + Conv3wCfg::TWO - 1) / Conv3wCfg::TWO) * ((Ho + Conv3wCfg::TH - 1) / Conv3wCfg::TH);
}

bool conv3w_supported(const GemmParams& p) {
    if (switches().no_conv3w || !conv3_supported(p)) return false;       // the same operand modes, epilogue, extents and weight copy as k_conv3.hip
    if (p.a.Cs % Conv3wCfg::KC) return false;                            // 32-channel maps: a pixel is 64 bytes and their rows are contiguous - k_conv3.hip's requests are whole lines there
    if (p.pool_out && conv3w_tiles(p) > (p.Mrows + kGemmBM - 1) / kGemmBM) return false;
    return true;
}

hipError_t launch_conv3w(const GemmParams& p, hipStream_t s) {
    using C = Conv3wCfg;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    static unsigned lds_ok = 0, lds_ok_pool = 0;   // per-device bits: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3w_kernel<false>, C::SMEM, lds_ok); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3w_kernel<true>, C::SMEM, lds_ok_pool); e != hipSuccess) return e;
    static int cus[32] = {0};     // compute units per device (one resident workgroup each)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    int ncu = __atomic_load_n(&cus[dev & 31], __ATOMIC_RELAXED);
    if (ncu == 0) {
        if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        __atomic_store_n(&cus[dev & 31], ncu, __ATOMIC_RELAXED);
    }
    const int tiles_x = (Wo + C::TWO - 1) / C::TWO, tiles_y = (Ho + C::TH - 1) / C::TH, nblk = p.N / 64;
    const int ntiles = p.B * tiles_x * tiles_y;
    const dim3 grid((unsigned)std::min(ntiles, ncu));
    if (p.pool_out) hipLaunchKernelGGL(conv3w_kernel<true>, grid, dim3(512), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, nblk, ntiles);
    else hipLaunchKernelGGL(conv3w_kernel<false>, grid, dim3(512), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, nblk, ntiles);
    return hipGetLastError();
}

}  // namespace w2x
