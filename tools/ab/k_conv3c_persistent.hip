// A/B record (not built into libw2x.so): persistent variant of csrc/k_conv3.hip's kernel - one workgroup of eight waves per CU walks
// over (tile, 64-channel block) items, halo tiles double-buffered by LDS-DMA, stores of a finished tile issued under the next
// products.  Bit-identical results; config 2 9.9-10.2 ms per frame against 9.8 ms for the two-workgroups-per-CU kernel that ships.
// Experiments (W2X_C3B_EXP, tools/ab/c3b_experiments.sh) on the 64 -> 64 layer at full resolution (691 GFLOP, 2.6 GB): 1.03 ms;
// weight fetches all hitting one line 1.03; no halo fetch 0.68; no stores 0.83; no products at all 0.74 - one 72 KB tile in flight
// per CU leaves the fetch latency-bound, and 160 KB of LDS has no room for a third buffer.
// Fragment of k_conv3.hip at that commit: needs its includes / typedefs (kernels.h, half8, float4v) to build.
// ---- second schedule: persistent workgroups, 16 x 62 output tiles, halo tiles by LDS-DMA into two buffers, weights through a
// register ring from L2 ----------------------------------------------------------------------------------------------------------
// conv3_kernel above moves BOTH operands of every MFMA through LDS, meets its workgroup at a barrier after every tap of every
// 32-channel chunk (16 MFMA per wave between barriers) and runs fetch -> products -> stores strictly one after the other in every
// workgroup; the two workgroups of a CU start together and stay in phase, so nothing overlaps (measured on cunet's layers, 460-620
// TFLOP/s: without the halo fetch -35 %, without the stores -25 %).  Here
//   * one workgroup of eight waves per CU walks over (tile, 64-channel block) items; the halo tile of the NEXT 32-channel chunk - or
//     of the next item's first chunk - is in flight into the second LDS buffer while the products of the current one run, and the
//     stores of a finished tile are issued after the next fetch has been started, under the next products;
//   * halo image in LDS: [18 rows][64 pixels][4 pieces of 8 channels], 64 bytes per pixel, no padding, the pieces of a pixel
//     rotated by 2 * ((x >> 2) & 3) slots.  One LDS-DMA instruction of a wave (buffer_load_dwordx4 ... lds: 16 bytes per lane,
//     lane-contiguous in LDS, no registers, no ds_write) moves 16 pixels - four lanes fetch the 64 contiguous bytes of a pixel, each
//     the piece that belongs in its slot.  A ds_read_b128 is served in groups of 16 lanes that pair the k-groups (g = 0 with 1, 2
//     with 3) over complementary row sets ({0-3, 12-15} with {4-11}: MI355X_MICROARCH.md, LDS) on 64 banks; of the four pixels
//     of a group that share x mod 4 two are read for piece g and two for g + 1, and the rotation sends them to four different
//     slots: conflict-free, where the padded tile of the first schedule (pixel stride 80 bytes) pays a second cycle on every group
//     (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).  The halo row is exactly 64 pixels, so a tile has 62 valid output
//     columns (same tile counts as 64 on cunet's extents); the two columns past them read the next row's first pixels and are
//     dropped.  Nothing is predicated: pixels past a ragged edge only feed outputs past it, the buffer resource ends with the tensor;
//   * a wave owns TWO adjacent output rows (2 x 64 pixels x 64 channels = 128 accumulator registers): a halo row's fragments are
//     read once and feed tap ky of the upper and tap ky - 1 of the lower row - 48 instead of 72 fragment reads per chunk;
//   * the weights never enter LDS: they are stored per 64-channel block and k-step as four contiguous fragments (fragorder.h
//     frag_conv3b) and stream from L2 through a ring of three taps (48 registers), each tap fetched one to two steps (32-48 MFMA)
//     before its first use;
//   * one barrier per chunk (288 MFMA per wave);
//   * the product is transposed (out^T = W X^T) with the A rows a permutation of the channels (row 4g + j of n-tile nt = channel
//     32 (nt >> 1) + 8g + 4 (nt & 1) + j of the block): a lane ends up with two runs of 8 consecutive channels of one pixel and a
//     store instruction writes 64 contiguous bytes per pixel, straight from registers.
struct Conv3cCfg {
    static constexpr int TH = 16, TWO = 62, HR = TH + 2, HP = 64, NWV = 8;
    static constexpr int ROWB = HP * 64;                               // bytes per halo row
    static constexpr int BUF = HR * ROWB + 256;                        // + the two-pixel over-read of the last row
    static constexpr int POOLB = NWV * 64 * 4;
    static constexpr int SMEM = 2 * BUF + POOLB;
};

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
constexpr unsigned kNoPix = 0xFFFFFFFFu;     // buffer offset past every resource: reads zeros, drops stores
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);   // raw buffer, 32-bit offsets, bounds-checked
}

struct C3Item { int b, oy0, ox0, nb, trem; };

template <bool POOL>
__global__ __launch_bounds__(512, 1) void conv3c_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, int nblk, int nitems) {
    using C = Conv3cCfg;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int Cin = p.a.Cs, nchunk = Cin / 32, KST = p.K / 32;
    const int tpi = tiles_x * tiles_y, G = gridDim.x;
    auto decode = [&](int item) {     // the blocks of a tile are neighbours in the item order: its halo comes from HBM once
        C3Item t;
        t.nb = item % nblk;
        const int tidx = item / nblk;
        t.b = tidx / tpi; t.trem = tidx - t.b * tpi;
        const int ty = t.trem / tiles_x, tx = t.trem - ty * tiles_x;
        t.oy0 = ty * C::TH; t.ox0 = tx * C::TWO;
        return t;
    };
    const size_t img_a = (size_t)p.a.Hs * p.a.Ws * Cin * 2, all_a = img_a * p.B;
    const unsigned rowb = (unsigned)p.a.Ws * (unsigned)Cin * 2u;
    // halo fetch of one chunk: wave wv < 4 moves (row, 16-pixel segment) = (q >> 2, q & 3) for q = 18 wv .. 18 wv + 17; lane = (pixel
    // lane >> 2 of the segment, slot lane & 3) fetches piece (slot - 2 ((x >> 2) & 3)) & 3, and (x >> 2) & 3 = lane >> 4
    const unsigned dma_vo = (unsigned)(lane >> 2) * (unsigned)Cin * 2u + (unsigned)(((lane & 3) - 2 * (lane >> 4)) & 3) * 16u;
    auto dma = [&](const C3Item& t, int kc, int buf) {
        const size_t org = (size_t)t.b * img_a + ((size_t)(p.a.y0 + t.oy0) * p.a.Ws + p.a.x0 + t.ox0) * Cin * 2;
        const size_t left = all_a - org;
        const __amdgpu_buffer_rsrc_t A = make_rsrc((const unsigned char*)p.a.p + org, left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)left);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int q = wv * 18 + i, row = q >> 2, seg = q & 3;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(A, (__attribute__((address_space(3))) void*)(smem + buf * C::BUF + row * C::ROWB + seg * 1024), 16,
                                                     dma_vo, (unsigned)row * rowb + (unsigned)(seg * 16 * Cin * 2 + kc * 64), 0, 0);
        }
    };
    const __amdgpu_buffer_rsrc_t W = make_rsrc(p.wt_perm, (unsigned)p.N * (unsigned)p.K * 2u);   // [N/64][KST][4][64][8]
    const unsigned wlane = lane * 16u;
    half8 w[3][4];                                                    // ring: slot = ky
    auto wload = [&](int slot, int nb, int kc, int ky, int kx, bool any) {   // !any: offset past the matrix, the fetch returns zeros nobody reads
#if W2X_C3B_EXP == 1
        const unsigned vo = any ? wlane + (unsigned)(nb * 0 + ky * 0 + kx * 0 + kc * 0) : kNoPix;
#else
        const unsigned vo = any ? wlane + (unsigned)(nb * KST + (ky * 3 + kx) * nchunk + kc) * 4096u : kNoPix;   // k-step (tap * Cin + 32 kc) / 32
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) w[slot][nt] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(W, __builtin_elementwise_add_sat(vo, nt * 1024u), 0, 0));
        asm volatile("" ::: "memory");                                // keeps the fetch where it is written
    };
    float4v acc[2][4][4];                                             // [row of the pair][16-pixel group][n-tile]
    auto mm = [&](int r, int slot, const half8 (&xa)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[r][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[slot][nt], xa[mt], acc[r][mt][nt], 0, 0, 0);
    };
    const float slope = p.act == 1 ? p.alpha : 1.f;                    // LeakyReLU as max(v, v * alpha), 0 <= alpha <= 1 (conv3_supported); 1: none
    const size_t img_o = (size_t)p.out.Hs * p.out.Ws * p.out.Cs * 2;
    auto epilogue = [&](const C3Item& t) {   // fp16, 2 x 16 bytes per lane and pixel; squeeze-excite partial sums of the stored values
        const __amdgpu_buffer_rsrc_t O = make_rsrc((unsigned char*)p.out.p + (size_t)t.b * img_o, (unsigned)img_o);
        const int n0 = t.nb * 64, xend = min(Wo, t.ox0 + C::TWO);
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
#if W2X_C3B_EXP == 3
                const bool valid = oy < Ho && ox < xend && acc[r][mt][0][0] == 12345.678f;
#else
                const bool valid = oy < Ho && ox < xend;
#endif
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
        if (POOL) {   // per-workgroup partial sums in a fixed order (pixels of a lane, lanes of a row group, waves 0..7); se_kernel adds the tiles of an image
            float* ws = (float*)(smem + 2 * C::BUF);                // [8][64]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = csum[nt][j];
                    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
                    if (fr == 0) ws[wv * 64 + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1) + j] = s;
                }
            __syncthreads();
            if (tid < 64) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < C::NWV; ++k) s += ws[k * 64 + tid];
                p.pool_out[((size_t)t.b * tpi + t.trem) * p.out.Cs + n0 + tid] = s;
            }
            __syncthreads();                                     // ws is free for the next tile
        }
    };

    int it = blockIdx.x, kc = 0, buf = 0;
    C3Item cur = decode(it), prev = cur;
    bool pend = false;
    if (wv < 4) dma(cur, 0, 0);
    wload(0, cur.nb, 0, 0, 0, true);
    wload(1, cur.nb, 0, 1, 0, true);
    // fragment reads: lane (fr, g) wants piece g of pixel x = kx + 16 mt + fr of a halo row
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = (kx + fr) * 64 + ((g + 2 * (((kx + fr) >> 2) & 3)) & 3) * 16;
    const unsigned char* xlane = smem + (2 * wv) * C::ROWB;
    for (;;) {
        const bool live = it < nitems;
        int nit = it, nkc = kc + 1;
        if (nkc == nchunk) { nkc = 0; nit = it + G; }
        const bool more = nit < nitems;
        const C3Item nxt = nkc == 0 ? decode(nit) : cur;
        if (live) {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's part of the halo tile has landed (and the ring's first taps, and last tile's stores)
            __syncthreads();                               // ... everybody's; everybody is done with the other buffer
            // Waves 0..3 fetch for all eight: a wave's memory operations return in order, so the weight fetches a wave issues after
            // its part of the halo fetch come back behind it (HBM latency instead of L2) and the wave stalls at its next tap; waves
            // i and i + 4 share a SIMD, so the matrix pipe belongs to the other wave meanwhile and stays busy
            if (more && wv < 4 && W2X_C3B_EXP != 2) dma(nxt, nkc, buf ^ 1);
        }
        if (kc == 0) {
            if (pend) epilogue(prev);
            if (!live) break;
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
        for (int kx = 0; kx < (W2X_C3B_EXP == 4 ? 0 : 3); ++kx) {
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
        if (kc == nchunk - 1) { prev = cur; pend = true; }
        it = nit; kc = nkc; cur = nxt; buf ^= 1;
    }
}

hipError_t launch_c3c(const GemmParams& p, int Ho, int Wo, hipStream_t s) {
    using C = Conv3cCfg;
    static unsigned lds_ok = 0, lds_ok_pool = 0;   // per-device bits: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3c_kernel<false>, C::SMEM, lds_ok); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3c_kernel<true>, C::SMEM, lds_ok_pool); e != hipSuccess) return e;
    static int cus[32] = {0};     // compute units per device (one resident workgroup each)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    int ncu = __atomic_load_n(&cus[dev & 31], __ATOMIC_RELAXED);
    if (ncu == 0) {
        if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        __atomic_store_n(&cus[dev & 31], ncu, __ATOMIC_RELAXED);
    }
    const int tiles_x = (Wo + C::TWO - 1) / C::TWO, tiles_y = (Ho + C::TH - 1) / C::TH, nblk = p.N / 64;
    const int nitems = p.B * tiles_x * tiles_y * nblk;
    const dim3 grid((unsigned)std::min(nitems, ncu));
    if (p.pool_out) hipLaunchKernelGGL(conv3c_kernel<true>, grid, dim3(512), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, nblk, nitems);
    else hipLaunchKernelGGL(conv3c_kernel<false>, grid, dim3(512), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, nblk, nitems);
    return hipGetLastError();
}

