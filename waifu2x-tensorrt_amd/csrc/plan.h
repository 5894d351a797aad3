// The "engine plan": what build() produces from an ONNX graph at a fixed [B,3,T,T] and what load() executes.
// It plays the role of the serialized TensorRT engine (/root/reference/src/tensorrt/img2img_build.cpp:142-161,
// img2img_load.cpp:158-203): a static sequence of fused HIP kernel launches over channel-last fp16 tensors.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace w2x {

// Activation tensor in HBM: [B][H][W][C] channel-last, fp16 (elt=2) or fp32 (elt=4, small side buffers).
struct TensorDesc {
    int B = 0, H = 0, W = 0, C = 0;  // stored dims (C may be padded with zero channels)
    int elt = 2;
    int64_t bytes() const { return (int64_t)B * H * W * C * elt; }
};

// A logical (possibly cropped) window of a stored tensor.
struct View {
    int t = -1;          // tensor index
    int H = 0, W = 0;    // logical extent
    int y0 = 0, x0 = 0;  // origin inside the stored tensor
};

enum Act : int { ACT_NONE = 0, ACT_LEAKY = 1, ACT_GELU = 2, ACT_RELU = 3, ACT_SIGMOID = 4 };
enum AMode : int { A_ROWS = 0, A_WIN = 1, A_CONV = 2 };
enum OMode : int { O_ROWS = 0, O_WIN = 1, O_PIXSHUF = 2 };

enum OpKind : int { OP_GEMM = 0, OP_ATTN = 1, OP_SE = 2, OP_SCALE_ADD = 3, OP_MLP = 4, OP_SWINATTN = 5 };

// Constant data blob (weights, tables) referenced by ops; uploaded once at load().
struct Blob {
    std::vector<uint8_t> data;
};

struct GemmOp {
    // A operand
    int amode = A_ROWS;
    View a;                 // source view; Cin = stored channels of tensor a.t
    int kh = 1, kw = 1, stride = 1;   // A_CONV taps
    int win_table = -1;     // blob: int32[H*W] window-order -> pixel index (A_WIN / O_WIN)
    int Mrows = 0;          // GEMM rows per batch item (logical output pixels of the A side)
    int aW = 0;             // width used to split a row index into (y,x) on the A side (= output W of the conv / rows view)
    int K = 0, N = 0;       // K = kh*kw*Cin (stored), N = logical output columns
    int w = -1;             // blob: fp16 W^T [N][K]
    int bias = -1;          // blob: fp32 [N] (already includes beta*W when ln)
    // LayerNorm folded algebraically: y = rstd*(x@W' - mean*csum) + bias'
    int ln = 0;
    int csum = -1;          // blob: fp32 [N]
    int stats_in = -1;      // tensor (fp32 [rows][2]: mean, rstd) produced by the op that wrote `a`
    // epilogue
    int act = ACT_NONE;
    float alpha = 0.f;      // leaky slope
    int has_clip = 0;
    float clip_lo = 0.f, clip_hi = 0.f;
    View res;               // optional residual (t = -1: none), same logical geometry as out
    View res2;              // optional second residual (cunet skip adds)
    int se_scale = -1;      // optional tensor fp32 [B][C]: per-(batch,channel) multiplier applied to A on load (cunet squeeze-excite gate folded into its consumer)
    int res_scale = -1;     // the same for the first residual operand
    int omode = O_ROWS;
    View out;
    int r = 1;              // O_PIXSHUF: upscale factor; N = r*r*Cout', columns ordered (dy,dx,c)
    int Cout = 0;           // logical channels per output pixel (N for O_ROWS/O_WIN, N/(r*r) for O_PIXSHUF)
    int stats_out = -1;     // tensor fp32 [rows][2] to fill with LayerNorm statistics of the produced rows, or -1
    float ln_eps = 1e-5f;   // eps used for stats_out
    int pool_out = -1;      // tensor fp32 [B][nblocks][C]: per-workgroup channel sums of the produced values (cunet SE squeeze)
};

struct AttnOp {
    int qkv = -1, out = -1;      // tensors: [rows][3C] and [rows][C] in window order
    int heads = 0, hd = 0, ws = 0;
    int nwin = 0;                // windows per batch item
    float scale = 1.f;
    int bias = -1;               // blob fp16 [nmask][heads][N][N] (relative position bias + shift mask)
    int maskid = -1;             // blob int32 [nwin]
    int nmask = 1;
};

// cunet squeeze-excite gate: s = sigmoid(W2 relu(W1 mean + b1) + b2) on [B][C] vectors.
struct SeOp {
    int pool = -1;     // tensor fp32 [B][nblocks][C]: per-workgroup partial sums written by the producing GEMM
    int nblocks = 0, Mrows = 0;
    int scale = -1;    // tensor fp32 [B][C] result
    int C = 0, Cmid = 0;
    float inv_count = 0.f;
    int w1 = -1, b1 = -1, w2 = -1, b2 = -1;  // blobs fp32
};

// fused Swin MLP branch (LayerNorm + fc1 + GELU + fc2 + residual) on contiguous token rows
struct MlpOp {
    int x = -1, y = -1;          // tensors [B][H][W][C]
    int C = 0;
    int w1 = -1, b1 = -1, w2 = -1, b2 = -1;   // blobs: fp16 [2C][C] (gamma folded), fp32 [2C] (beta folded), fp16 [C][2C], fp32 [C]
    float eps = 1e-5f;
    int stats_out = -1; float eps_out = 1e-5f;
};

// fused Swin attention branch (LayerNorm + shift/partition + qkv + W-MSA + proj + reverse + residual)
struct SwinAttnOp {
    int x = -1, y = -1;
    int C = 0, heads = 0, hd = 0, ws = 0, nwin = 0;
    int table = -1;                 // blob int32[H*W]
    int H = 0, W = 0, ry = -1, rx = -1;   // if ry >= 0 the table equals pixel = ((y+ry)%H)*W + (x+rx)%W of the window-order (y,x): computed in-kernel
    int wqkv = -1, bqkv = -1, wproj = -1, bproj = -1, bias = -1, maskid = -1;
    float scale = 1.f, eps = 1e-5f;
    int stats_out = -1; float eps_out = 1e-5f;
};

struct Op {
    int kind = OP_GEMM;
    std::string name;
    GemmOp g;
    AttnOp at;
    SeOp se;
    MlpOp m;
    SwinAttnOp sa;
    double flops = 0;  // algorithmic 2*MACs of the ONNX nodes this op covers
};

struct Plan {
    int B = 0, Cin = 3, T = 0;     // network input [B,3,T,T]; B = tiles per network pass (userB x super-batch factor)
    int userB = 0;                 // RenderConfig::batchSize the plan was built for (B is a multiple of it)
    int Tout = 0, Cout = 3;        // network output [B,3,Tout,Tout]
    int elt = 2;                   // bytes per activation / weight element: 2 = fp16 (Precision::FP16), 4 = fp32 (Precision::TF32 requests)
    int in_tensor = -1, out_tensor = -1;
    std::vector<TensorDesc> tensors;
    std::vector<Blob> blobs;
    std::vector<Op> ops;
    double flops = 0;              // sum of op flops (per batch of B tiles)
    std::string model_kind;        // "swin_unet" / "cunet" / "generic"
    std::string describe() const;
    // binary (de)serialization for the on-disk engine file
    std::vector<uint8_t> serialize() const;
    static Plan deserialize(const uint8_t* p, size_t n);   // throws on a truncated or inconsistent file (validate())
    void validate() const;
};

}  // namespace w2x
