// Launch interface of the HIP kernels (gfx950).  Plain C structs, no torch types.
#pragma once
#include <hip/hip_runtime.h>

#include "support.h"
#include "switches.h"
#include <cstdint>

namespace w2x {

constexpr int kGemmBM = 128;   // rows per GEMM workgroup tile (every instantiation)

// The opt-in to more than 64 KiB of dynamic LDS is a per-device attribute of a kernel: every launcher calls this with its own
// static mask before launching, so that engines on several devices of one process (one host thread each) all get it.
inline hipError_t ensure_dynamic_lds(const void* fn, int bytes, unsigned& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned bit = 1u << (dev & 31);
    if (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
    return e;
}

// Cache policy of the kernels' OUTPUT stores (round 6).  What a launch produces is read by the next launch, never by itself: with the nt (streaming) policy the rows
// stay out of L2, which they would only pass through, and leave it to the weights and to the rows a launch fetches a second time.  The four transformer kernels together:
// -0.9 % per frame in five alternating runs on two boxes (profiles/r6_kernels/lib_nt_stores_frame_level.txt).  W2X_ST_AUX = the aux operand of the buffer stores
// (2 = nt, 0 = default policy; per file: tools/ab/lib_variants.sh "<file>:-DW2X_ST_AUX=0"); w2x_store_out() is the same choice for stores through a plain pointer.
#ifndef W2X_ST_AUX
#define W2X_ST_AUX 2
#endif
// cache policy of a kernel's LAST read of its input rows (the residual fetch of the transformer kernels; the C = 96 MLP reads its rows once)
#ifndef W2X_LD_LAST_AUX
#define W2X_LD_LAST_AUX 2
#endif
#ifdef __HIPCC__
template <class T> __device__ __forceinline__ void w2x_store_out(T* p, const T v) {
#if W2X_ST_AUX == 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#endif

struct TView {           // device view of a channel-last tensor
    void* p = nullptr;
    int Hs = 0, Ws = 0, Cs = 0;  // stored dims
    int y0 = 0, x0 = 0;          // origin of the logical window
};

struct GemmParams {
    TView a;
    int amode = 0, kh = 1, kw = 1, stride = 1;
    int B = 0, Mrows = 0, aW = 0;
    const int* win_table = nullptr;
    int K = 0, N = 0, Kw = 0;       // Kw: row stride of wt (K rounded up to 8, zero padded)
    const void* wt = nullptr;       // fp16 [N][Kw]
    const void* wt_frag = nullptr;  // fragment-major copy (fragorder.h) for k_pixgemm.hip, or null
    const void* wt_perm = nullptr;  // fragorder.h frag_conv3b copy for k_conv3.hip's conv3b_kernel, or null
    const float* bias = nullptr;    // [N]
    const float* csum = nullptr;    // [N] (ln)
    const float* stats_in = nullptr;  // [pixels of a][2]
    int ln = 0;
    int act = 0; float alpha = 0.f;
    int has_clip = 0; float clip_lo = 0.f, clip_hi = 0.f;
    const float* a_scale = nullptr;   // fp32 [B][a.Cs]: squeeze-excite gate of the input map, applied to A on load (rounded to fp16 like the in-place pass)
    const float* res_scale = nullptr; // fp32 [B][res.Cs]: the same for the first residual
    TView res, res2;                // p == nullptr: none
    TView out;
    int omode = 0, r = 1, Cout = 0;
    float* stats_out = nullptr; float ln_eps = 1e-5f;
    float* pool_out = nullptr;      // [B][ceil(Mrows/kGemmBM)][out.Cs] per-workgroup column sums (squeeze-excite)
};

struct AttnParams {
    const void* qkv = nullptr; void* out = nullptr;
    int B = 0, nwin = 0, heads = 0, hd = 0, ntok = 0;
    float scale = 1.f;
    const void* bias = nullptr;     // fp16 [nmask][heads][ntok][ntok]
    const int* maskid = nullptr;    // [nwin]
};

struct MlpParams {               // y = x + W2 gelu(W1 LN(x) + b1) + b2 on contiguous rows [M][C]
    const void* x = nullptr; void* y = nullptr;
    long M = 0; int C = 0;
    const void* w1 = nullptr;      // fp16 [2C][C], LayerNorm gamma folded in
    const float* b1 = nullptr;     // [2C], LayerNorm beta folded in
    const void* w2 = nullptr;      // fp16 [C][2C]
    const float* b2 = nullptr;     // [C]
    float eps = 1e-5f;
    float* stats_out = nullptr; float eps_out = 1e-5f;
    // fragment-major copies for k_mlp2.hip (engine.cpp): w1 as [hidden tile][k-step][lane][8]; w2 as
    // [chunk of 32 hidden][n-tile][lane][8] with the k order of the GELU'd accumulators (slots 0..3: hidden 4g+j, 4..7: 16+4g+j)
    const void* w1_frag = nullptr; const void* w2_frag = nullptr;
    // true: the fragment copies are in the 32x32x16 order of k_mlp96q.hip (fragorder.h frag32_major / frag32_w2; C = 96 in the engine)
    bool frag32 = false;
    // Image head folded into the launch (engine.cpp: the plan's last MLP followed by Linear 96 -> 64 = 4x4 sub-pixels x 4 stored channels, DepthToSpace(4), Clip):
    // y is not stored; every produced row goes through the head instead, out[b][4 oy + dy][4 ox + dx][0..3] = clip(fp16(y_row W^T + bias)) - the arithmetic of
    // toimage_kernel (k_pixgemm.hip), bit for bit.  ti_w = null: no head.
    const void* ti_w = nullptr;    // fp16 fragment-major [4 n-tiles][3 k-steps][64 lanes][8] (the head's GemmParams::wt_frag)
    const float* ti_b = nullptr;   // fp32 [64]
    void* ti_out = nullptr; int ti_Hs = 0, ti_Ws = 0, ti_Mrows = 0, ti_aW = 0;   // output map [B][Hs][Ws][4], rows per image, row width of the token map
    int ti_clip = 0; float ti_lo = 0.f, ti_hi = 0.f;
    long ti_row0 = 0;              // (set by the launcher when it cuts a pass into runs: global index of this run's first row)
};

struct SwinAttnParams {          // y = x + proj(W-MSA(LN(x))) on token maps [B][H][W][C], window 6x6
    const void* x = nullptr; void* y = nullptr;
    const int* table = nullptr;    // int32[H*W]: window-order row -> pixel (shift + partition); also the scatter map
    int H = 0, W = 0, ry = -1, rx = -1;   // ry >= 0: closed form of the table, pixel = ((y+ry)%H)*W + (x+rx)%W (no lookup)
    int B = 0, nwin = 0, C = 0, hd = 0;
    const void* wqkv = nullptr;    // fp16 [3C][C], LayerNorm gamma folded in
    const float* bqkv = nullptr;   // [3C], LayerNorm beta folded in
    float scale = 1.f;
    const float* bias32 = nullptr; // fp32 [nmask][heads][3][576]: (rel-pos bias + shift mask) * log2(e) in the kernel's lane order (lower.cpp)
    const int* maskid = nullptr;   // [nwin]
    const void* wproj = nullptr;   // fp16 [C][C]
    const float* bproj = nullptr;  // [C]
    float eps = 1e-5f;
    float* stats_out = nullptr; float eps_out = 1e-5f;
    // the same two matrices in MFMA-fragment order (engine.cpp frag_major): [16-row tile][32-column k-step][lane][8], so a
    // wave's fragment load is one contiguous KiB.  Required by k_swinattn96.hip / k_swinattn192u.hip.
    const void* wqkv_frag = nullptr; const void* wproj_frag = nullptr;
};

struct SeParams {
    const float* pool = nullptr; float* scale = nullptr;   // pool: [B][nblocks][Cs] partial sums from the producing GEMM (nblocks row tiles per batch item)
    int B = 0, C = 0, Cs = 0, Cmid = 0; float inv_count = 0.f;
    int nblocks = 0, Mrows = 0;
    const float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
};

struct TileSlot { int x, y, aug, valid; };   // input rect origin (may be negative), augmentation 0..7, 0 = zero-pad slot

struct GatherParams {
    const uint8_t* frame = nullptr; int rows = 0, cols = 0; size_t step = 0;  // u8 BGR interleaved (deep: u16 samples; step in bytes)
    int deep = 0;
    void* out = nullptr;            // fp16 (fp32 engines: fp32) [B][T][T][4]
    int fp32 = 0;
    const TileSlot* slots = nullptr;
    int B = 0, T = 0;
};

struct ComposeParams {
    const void* tiles = nullptr;    // fp16 (fp32 engines: fp32) [slots][To][To][4], slot = tile*steps + aug
    int fp32 = 0;
    uint8_t* dst = nullptr; size_t dst_step = 0;    // u8 BGR (deep: u16 samples; step in bytes)
    int deep = 0;
    int outW = 0, outH = 0;
    int To = 0;
    int nx = 0, ny = 0;
    int stride_x = 0, stride_y = 0;   // To - outOverlap
    int ovx = 0, ovy = 0;             // blend ramp lengths (0: no blending)
    const float* ramp_x = nullptr;    // [ovx] left ramp, fl32(double(i+1)/(ovx+1))
    const float* ramp_y = nullptr;
    int tta = 0;
    int tta_bug_compat = 0;
    // strip rendering (one GPU of several composes only its columns): output columns [x0, x1) and the global index of
    // the tile held by slot 0 (x1 = 0: the whole canvas)
    int x0 = 0, x1 = 0; long first_tile = 0;
    // shard rendering (renderSharded): additionally output rows [y0, y1) (y1 = 0: all rows)
    int y0 = 0, y1 = 0;
};

hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
bool pixgemm_supported(const GemmParams& p);                    // k_pixgemm.hip: streaming kernel for pixel-shuffle projections
hipError_t launch_pixgemm(const GemmParams& p, hipStream_t s);
bool conv3_supported(const GemmParams& p);                      // k_conv3.hip: LDS-tiled direct 3x3 convolution
// k_stem.hip: first 3x3 convolution on the 4-halves-per-pixel input tile (no LDS, weights in registers)
bool stem_supported(const GemmParams& p);
hipError_t launch_stem(const GemmParams& p, hipStream_t s);
hipError_t launch_conv3(const GemmParams& p, hipStream_t s);
bool conv3h_supported(const GemmParams& p);                     // k_conv3h.hip: direct 3x3 convolution onto 4 stored channels (cunet's image heads)
hipError_t launch_conv3h(const GemmParams& p, hipStream_t s);
bool conv48_supported(const GemmParams& p);                     // k_conv48.hip: direct 3x3 convolution 48 -> 96 channels (swin_unet's patch convolution)
hipError_t launch_conv48(const GemmParams& p, hipStream_t s);
bool conv48_stem_supported(const GemmParams& p, const GemmParams& ps);   // k_conv48.hip: the stem launch ps folded into the patch convolution p that alone reads its output
hipError_t launch_conv48_stem(const GemmParams& p, const GemmParams& ps, hipStream_t s);
bool conv3_stem_supported(const GemmParams& p, const GemmParams& ps);    // k_conv3.hip: the same for cunet's stem (4 -> 32) in front of its 32 -> 64 convolution
hipError_t launch_conv3_stem(const GemmParams& p, const GemmParams& ps, hipStream_t s);
bool conv3_up_supported(const GemmParams& p, const GemmParams& q);       // k_conv3.hip: cunet's ConvTranspose (pixel-shuffle projection q, LeakyReLU, skip add) computed in the halo stage of the 64 -> 64 convolution p behind it
hipError_t launch_conv3_up(const GemmParams& p, const GemmParams& q, hipStream_t s);
int conv3_tiles(const GemmParams& p);                           // workgroups (= pooling partials) per image of launch_conv3
hipError_t launch_attn(const AttnParams& p, hipStream_t s);
// k_f32.hip: the fp32 engine's kernels (Plan::elt == 4): general GEMM / convolution and the window attention core on fp32 rows
hipError_t launch_gemm_f32(const GemmParams& p, hipStream_t s, bool exact);      // exact: fp32 products (Precision::FP32); else three bf16 products per k-step (Precision::TF32)
hipError_t launch_attn_f32(const AttnParams& p, hipStream_t s);
// Fused MLP branch on fp32 rows with split-bf16 products (Precision::TF32; k_f32.hip mlp32_kernel): y = x + W2 gelu(W1 ((x - mean) rstd) + b1) + b2, the row statistics
// from the producer's stats tensor.  w1h / w1l, w2h / w2l: the bf16 hi / lo planes of W1 [2C][C] (LayerNorm gamma folded) and W2 [C][2C] in fragment-major order
// (fragorder.h frag_major of each plane).  The engine launches it in place of an fc1 / fc2 pair of gemm32 launches.
struct Mlp32Params {
    const float* x = nullptr; float* y = nullptr;
    long M = 0; int C = 0;
    const float* stats_in = nullptr;     // [M][2]: mean, rstd
    const void *w1h = nullptr, *w1l = nullptr, *w2h = nullptr, *w2l = nullptr;
    const float *b1 = nullptr, *b2 = nullptr;
    float* stats_out = nullptr; float eps_out = 1e-5f;
};
bool mlp32_supported(int C);
hipError_t launch_mlp32(const Mlp32Params& p, hipStream_t s);
// Fused Swin attention branch on fp32 rows with split-bf16 products (Precision::TF32; k_f32.hip swinattn32_kernel): y = res + proj(W-MSA((x - mean) rstd)), in place of the
// un-fused plan's qkv gemm -> attention core -> proj gemm.  Rows are pixels of plain [pixels][C] maps; table_in / table_out: window-order row -> pixel inside a batch item
// (the qkv op's gather table and the proj op's scatter table); weights as bf16 hi / lo planes, fragment-major (frag_major of each plane of W [3C][C] / [C][C]).
struct SwinAttn32Params {
    const float* x = nullptr; float* y = nullptr; const float* res = nullptr;
    int B = 0, nwin = 0, C = 0;
    long pix_per_item = 0;
    const int* table_in = nullptr; const int* table_out = nullptr;
    const float* stats_in = nullptr;
    const void *wqkv_h = nullptr, *wqkv_l = nullptr, *wproj_h = nullptr, *wproj_l = nullptr;
    const float *bqkv = nullptr, *bproj = nullptr;
    float scale = 1.f;
    const float* bias = nullptr; const int* maskid = nullptr;      // fp32 [nmask][6][36][36], [nwin]
    float* stats_out = nullptr; float eps_out = 1e-5f;
};
bool swinattn32_supported(int C, int heads, int hd, int ntok);
hipError_t launch_swinattn32(const SwinAttn32Params& p, hipStream_t s);
hipError_t launch_mlp(const MlpParams& p, hipStream_t s);
hipError_t launch_swin_attn(const SwinAttnParams& p, hipStream_t s);
// swin_attn_supported / mlp_supported / gemm_row_stats_supported / attn_supported: support.h
hipError_t launch_se(const SeParams& p, hipStream_t s);
hipError_t launch_scale(void* x, const float* scale, int B, int HW, int Cs, bool fp32, hipStream_t s);
hipError_t launch_gather(const GatherParams& p, hipStream_t s);
hipError_t launch_compose(const ComposeParams& p, hipStream_t s);
// debug/test helpers used by w2x_infer (mirrors blobFromImages / imagesFromBlob, img2img_infer.cpp:5-39)
hipError_t launch_blob_to_nhwc(const float* nchw, void* out_nhwc4, int B, int T, bool fp32, hipStream_t s);
hipError_t launch_nhwc_to_blob(const void* in_nhwc4, float* nchw, int B, int T, bool fp32, hipStream_t s);

#ifdef __HIPCC__
namespace gate {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// x * s on 8 halves, s fp32 per channel (cunet's squeeze-excite gates: the in-place pass scale_kernel and the gated operand loads of
// k_gemm / k_pixgemm all go through this, so folding a gate never changes a bit): v_fma_mixlo / mixhi read the f16 halves directly,
// multiply by the fp32 gate and write f16 - one instruction per element
typedef unsigned gate_u4 __attribute__((ext_vector_type(4)));
typedef float gate_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 gate8(const half8 v, const float* sc) {
    const gate_f4 s0 = *(const gate_f4*)sc, s1 = *(const gate_f4*)(sc + 4);
    const float s[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    gate_u4 x = __builtin_bit_cast(gate_u4, v), o;
    const float zero = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned r;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x[d]), "v"(s[2 * d]), "v"(zero));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(x[d]), "v"(s[2 * d + 1]), "v"(zero));
        o[d] = r;
    }
    return __builtin_bit_cast(half8, o);
}
}  // namespace gate
#endif

}  // namespace w2x
