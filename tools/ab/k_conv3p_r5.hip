// Direct 3x3 convolution (stride 1, valid) for gfx950 - the PERSISTENT form of k_conv3.hip (same products in the same order, same weight copy):
//     out[b][y][x][n] = act( sum_{ky,kx,c} in[b][y+ky][x+kx][c] * W[n][(ky*3+kx)*Cin + c] + bias[n] )
//
// k_conv3.hip runs fetch -> products -> stores one after the other in every workgroup, and the two workgroups of a CU stay in phase: its products
// alone would take 0.63 of its time.  Here ONE workgroup of EIGHT waves per CU walks over (tile, 64-channel block) items:
//   * tile = 16 rows x 64 columns; halo image of a 32-channel chunk 18 rows x 66 pixels x 64 bytes = 74.25 KB, TWO of them in LDS (+ 4 KB of pooling
//     partials = 152.5 KB): the next chunk's image - or the next item's first - is in flight by LDS-DMA (buffer_load ... lds: no registers, no ds_write;
//     four lanes fetch the 64 bytes of a pixel's chunk) while the products of the current one run, and the stores of a finished tile complete under the
//     next products (only the fetches are waited for: s_waitcnt before the stores are issued);
//   * wave v owns output rows 2 v, 2 v + 1: k_conv3.hip's inner loop unchanged (128 accumulator registers, a halo row's fragments feed tap ky of the
//     upper and ky - 1 of the lower row, weights from L2 through a ring of three taps), so every output is the same fp32 sum in the same order;
//   * one barrier per chunk (288 MFMA per wave between two);
//   * halo image in LDS as in k_conv3.hip: pixel P = row * 66 + x at byte 64 P, its four 16-byte pieces rotated by 2 ((x >> 2) & 3) slots (conflict-free
//     ds_read_b128).  The DMA writes lane-contiguously, so a lane fetches the piece that belongs in its slot.
// The squeeze-excite partials are per tile (16 rows here, 8 there): the pooled means differ from k_conv3.hip's in the order of their fp32 sums.
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

struct Conv3pCfg {
    static constexpr int NWV = 8, TH = 2 * NWV, TW = 64, HR = TH + 2, HC = TW + 2;
    static constexpr int ROWB = HC * 64;                                // bytes per halo row (32 channels)
    static constexpr int BUF = HR * ROWB;                               // 76 032
    static constexpr int NDMA = (BUF + 1023) / 1024;                    // wave instructions per image (75; the last one a quarter full)
    static constexpr int NQ = (NDMA + NWV - 1) / NWV;                   // per wave
    static constexpr int POOLB = 2 * NWV * 64 * 4;                      // pooling partials [2 tiles in flight][8 row pairs][64 channels] floats
    static constexpr int SMEM = 2 * BUF + POOLB;                        // 156 160 <= 163 840
};
struct C3pItem { int b, oy0, ox0, nb, tidx; };

// Diagnostic build only (-DW2X_C3P_STAMPS): s_memtime deltas per phase, one record per wave, plain stores.  0 barrier at the top of a chunk, 1 DMA issue,
// 2 epilogue, 3 bias + products + wait for the fetches, 4 chunks, 5 waves
#ifdef W2X_C3P_STAMPS
__device__ unsigned long long g_c3p_stamps[256 * 8][8];
#define W2X_STAMP(K) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); tacc[K] += t_ - tprev; tprev = t_; }
#else
#define W2X_STAMP(K)
#endif

template <bool POOL>
__global__ __launch_bounds__(512, 1) void conv3p_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, int nblk, int ntiles) {
    using C = Conv3pCfg;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int Cin = p.a.Cs, nchunk = Cin / 32, KST = p.K / 32;
    const int tpi = tiles_x * tiles_y, G = gridDim.x;
    // this workgroup's items: all blocks of tile blockIdx.x, then of tile blockIdx.x + G, ... (a tile's blocks re-read its halo through this XCD's L2)
    auto decode = [&](int c) {
        C3pItem t;
        t.nb = c % nblk;
        t.tidx = (int)blockIdx.x + (c / nblk) * G;
        t.b = t.tidx / tpi;
        const int trem = t.tidx - t.b * tpi, ty = trem / tiles_x, tx = trem - ty * tiles_x;
        t.oy0 = ty * C::TH; t.ox0 = tx * C::TW;
        return t;
    };
    const int my_items = blockIdx.x < (unsigned)ntiles ? ((ntiles - 1 - (int)blockIdx.x) / G + 1) * nblk : 0;
    if (my_items == 0) return;
    const size_t img_a = (size_t)p.a.Hs * p.a.Ws * Cin * 2, all_a = img_a * p.B;
    const unsigned rowb = (unsigned)p.a.Ws * (unsigned)Cin * 2u;
    // halo fetch of one chunk: instruction q (of NDMA; wave v issues q = v, v + 8, ...) fills LDS bytes [1024 q, 1024 q + 1024) = pixels 16 q .. 16 q + 15; lane =
    // (pixel lane >> 2, slot lane & 3) fetches piece (slot - rot(x)) & 3 of pixel P = 16 q + (lane >> 2) = (row P / 66, x P % 66).  The lane's part of the address
    // is worked out per instruction: a dozen integer operations, against twenty registers to keep it.
    auto dma = [&](const C3pItem& t, int kc, int buf) {
        const size_t org = (size_t)t.b * img_a + ((size_t)(p.a.y0 + t.oy0) * p.a.Ws + p.a.x0 + t.ox0) * Cin * 2;
        const size_t left = all_a - org;
        const __amdgpu_buffer_rsrc_t A = make_rsrc((const unsigned char*)p.a.p + org, left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)left);
        const int hrows = min(C::HR, Ho + 2 - t.oy0), hcols = min(C::HC, Wo + 2 - t.ox0);
#pragma unroll
        for (int i = 0; i < C::NQ; ++i) {
            const int q = wv + C::NWV * i;
            const int P = 16 * q + (lane >> 2), row = P / C::HC, x = P - row * C::HC;
            const int piece = ((lane & 3) - 2 * ((x >> 2) & 3)) & 3;
            const unsigned vo = (row < hrows && x < hcols) ? (unsigned)row * rowb + (unsigned)(x * Cin * 2 + piece * 16) : kNoPix;      // outside the map: zeros
            if (q < C::NDMA && row < C::HR)                                           // (past the image: the lanes stay off, they would write into the other buffer)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(A, (__attribute__((address_space(3))) void*)(smem + buf * C::BUF + q * 1024), 16, vo, (unsigned)(kc * 64), 0, 0);
        }
    };
    const __amdgpu_buffer_rsrc_t W = make_rsrc(p.wt_perm, (unsigned)p.N * (unsigned)p.K * 2u);   // [N/64][KST][4][64][8]
    const unsigned wlane = lane * 16u;
    half8 w[3][4];                                                      // ring: slot = ky
    auto wload = [&](int slot, int nb, int kc, int ky, int kx, bool any) {   // !any: offset past the matrix, the fetch returns zeros nobody reads
        const unsigned vo = any ? wlane + (unsigned)(nb * KST + (ky * 3 + kx) * nchunk + kc) * 4096u : kNoPix;   // k-step (tap * Cin + 32 kc) / 32 of block nb
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) w[slot][nt] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(W, __builtin_elementwise_add_sat(vo, nt * 1024u), 0, 0));
        asm volatile("" ::: "memory");                                  // keeps the fetch where it is written
    };
    float4v acc[2][4][4];                                               // [row of the pair][16-pixel group][n-tile]
    auto mm = [&](int r, int slot, const half8 (&xa)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[slot][nt], xa[mt], acc[r][mt][nt], 0, 0, 0);
    };
    const float slope = p.act == 1 ? p.alpha : 1.f;                    // LeakyReLU as max(v, v * alpha), 0 <= alpha <= 1 (conv3_supported); 1: none
    const size_t img_o = (size_t)p.out.Hs * p.out.Ws * p.out.Cs * 2;
    auto epilogue = [&](const C3pItem& t, int seq) {   // fp16, 2 x 16 bytes per lane and pixel; squeeze-excite partial sums of the stored values
        const __amdgpu_buffer_rsrc_t O = make_rsrc((unsigned char*)p.out.p + (size_t)t.b * img_o, (unsigned)img_o);
        const int n0 = t.nb * 64;
        float csum[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[nt][j] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = t.oy0 + 2 * wv + r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ox = t.ox0 + mt * 16 + fr;
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
        if (POOL) {   // per-tile partial sums in a fixed order (pixels of a lane, lanes of a row group, then row pairs 0..7 in pool_reduce)
            float* ws = (float*)(smem + 2 * C::BUF) + (seq & 1) * (C::NWV * 64);   // [8 row pairs][64 channels of the block], two tiles in flight
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sum = csum[nt][j];
                    sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
                    if (fr == 0) ws[wv * 64 + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1) + j] = sum;
                }
        }
    };
    // the partials of item `seq` are complete once every wave has run its epilogue and a barrier has passed; wave 0 adds the row pairs
    auto pool_reduce = [&](const C3pItem& t, int seq) {
        if (POOL && wv == 0) {
            const float* ws = (const float*)(smem + 2 * C::BUF) + (seq & 1) * (C::NWV * 64);
            float s = ws[lane];
#pragma unroll
            for (int v = 1; v < C::NWV; ++v) s += ws[v * 64 + lane];
            p.pool_out[(size_t)t.tidx * p.out.Cs + t.nb * 64 + lane] = s;
        }
    };

#ifdef W2X_C3P_STAMPS
    unsigned long long tprev, tacc[5] = {};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
    int c = 0, kc = 0, buf = 0, seq = 0;                 // seq: tiles finished
    C3pItem cur = decode(0), red = cur;
    bool red_pending = false;
    dma(cur, 0, 0);
    wload(0, cur.nb, 0, 0, 0, true);
    wload(1, cur.nb, 0, 1, 0, true);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    // fragment reads: lane (fr, g) wants piece g of pixel x = kx + 16 mt + fr of a halo row ((x >> 2) & 3 does not depend on mt)
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = (kx + fr) * 64 + ((g + 2 * (((kx + fr) >> 2) & 3)) & 3) * 16;
    const unsigned char* xlane = smem + (2 * wv) * C::ROWB;
    for (;;) {
        int nc = c, nkc = kc + 1;
        if (nkc == nchunk) { nkc = 0; nc = c + 1; }
        const bool more = nc < my_items;
        const C3pItem nxt = nkc == 0 ? (more ? decode(nc) : cur) : cur;
        __syncthreads();                                   // everybody's part of this chunk's image has landed (each wave waited for its own); everybody is done with the other buffer
        W2X_STAMP(0)
        if (red_pending) { pool_reduce(red, seq - 1); red_pending = false; }
        if (more) dma(nxt, nkc, buf ^ 1);
        W2X_STAMP(1)
        if (kc == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4v bv = *(const float4v*)(p.bias + cur.nb * 64 + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1));
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = bv;
            }
        }
        const unsigned char* xb = xlane + buf * C::BUF;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            half8 xa[4];
            auto xload = [&](int hh) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(xb + xoff[kx] + hh * C::ROWB + mt * 1024);
            };
            xload(0);
            mm(0, 0, xa);
            xload(1);
            wload(2, cur.nb, kc, 2, kx, true);
            mm(0, 1, xa); mm(1, 0, xa);
            xload(2);
            if (kx < 2) wload(0, cur.nb, kc, 0, kx + 1, true); else wload(0, nxt.nb, nkc, 0, 0, more);
            mm(0, 2, xa); mm(1, 1, xa);
            xload(3);
            if (kx < 2) wload(1, cur.nb, kc, 1, kx + 1, true); else wload(1, nxt.nb, nkc, 1, 0, more);
            mm(1, 2, xa);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's part of the next image (and the ring's first taps) - BEFORE the stores go out, which nobody waits for
        W2X_STAMP(3)
#ifdef W2X_C3P_STAMPS
        tacc[4] += 1;
#endif
        if (kc == nchunk - 1) { epilogue(cur, seq); red = cur; red_pending = true; ++seq; W2X_STAMP(2) }
        if (!more) break;
        c = nc; kc = nkc; cur = nxt; buf ^= 1;
    }
#ifdef W2X_C3P_STAMPS
    if (lane == 0 && blockIdx.x < 256) { for (int k = 0; k < 5; ++k) g_c3p_stamps[blockIdx.x * 8 + wv][k] += tacc[k]; g_c3p_stamps[blockIdx.x * 8 + wv][5] += 1; }
#endif
    if (POOL) {      // the last tile's partials
        __syncthreads();
        if (red_pending) pool_reduce(red, seq - 1);
    }
}

}  // namespace

#ifdef W2X_C3P_STAMPS
// out[6]: phase sums over all waves, then cleared
extern "C" void w2x_c3p_stamps(unsigned long long* out) {
    static unsigned long long h[256 * 8][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3p_stamps), sizeof(h));
    for (int k = 0; k < 6; ++k) out[k] = 0;
    for (int w = 0; w < 256 * 8; ++w) for (int k = 0; k < 6; ++k) out[k] += h[w][k];
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c3p_stamps), h, sizeof(h));
}
#endif

int conv3p_tiles(const GemmParams& p) {   // pooling partials per image and 64-channel block
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    return ((Wo + Conv3pCfg::TW - 1) / Conv3pCfg::TW) * ((Ho + Conv3pCfg::TH - 1) / Conv3pCfg::TH);
}

bool conv3p_supported(const GemmParams& p) {
    if (switches().no_conv3p || !conv3_supported(p)) return false;       // the same operand modes, epilogue, extents and weight copy as k_conv3.hip (and fewer pooling partials)
    return (size_t)(Conv3pCfg::HR + 1) * p.a.Ws * p.a.Cs * 2 <= 0x7FFFFFFFull;     // 32-bit offsets inside a halo image
}

hipError_t launch_conv3p(const GemmParams& p, hipStream_t s) {
    using C = Conv3pCfg;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    static unsigned lds_ok = 0, lds_ok_pool = 0;   // per-device bits: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3p_kernel<false>, C::SMEM, lds_ok); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3p_kernel<true>, C::SMEM, lds_ok_pool); e != hipSuccess) return e;
    static int cus[32] = {0};     // compute units per device (one resident workgroup each)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    int ncu = __atomic_load_n(&cus[dev & 31], __ATOMIC_RELAXED);
    if (ncu == 0) {
        if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        __atomic_store_n(&cus[dev & 31], ncu, __ATOMIC_RELAXED);
    }
    const int tiles_x = (Wo + C::TW - 1) / C::TW, tiles_y = (Ho + C::TH - 1) / C::TH, nblk = p.N / 64;
    const int ntiles = p.B * tiles_x * tiles_y;
    const dim3 grid((unsigned)std::min(ntiles, ncu));
    if (p.pool_out) hipLaunchKernelGGL(conv3p_kernel<true>, grid, dim3(512), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, nblk, ntiles);
    else hipLaunchKernelGGL(conv3p_kernel<false>, grid, dim3(512), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, nblk, ntiles);
    return hipGetLastError();
}

}  // namespace w2x
