// Pre/post kernels of the tile pipeline (HBM-bound byte/fp32 work, bit-exact against the oracle):
//   gather  : frame u8 BGR -> network input tiles fp16 [B][T][T][4] (RGB, x*fl32(1/255)), replicate padding and the
//             TTA dihedral transform folded into the source index.   Replaces cv::cuda::cvtColor(BGR2RGB)
//             (img2img_render.cpp:227), padRoi (:68-105), applyAugmentation (:134-177), blobFromImages
//             (img2img_infer.cpp:5-21) and the D2D copy (:76-77).
//   compose : network output tiles fp16 -> frame u8 BGR.  Per output pixel, the covering tiles are visited in
//             ascending tile index, TTA de-augmentation + fp32 sum in aug order + *0.125f, ramp weights multiplied
//             in the order left, top, right, bottom, fp32 overlap-add, *255 -> rint -> saturate -> BGR.  Replaces
//             imagesFromBlob (img2img_infer.cpp:23-39), reverseAugmentation/TTA accumulate (:179-222,:305-318),
//             applyWeights (:107-121), the canvas add (:329-330) and the final convertTo/cvtColor (:342-343).
//   se/scale: cunet squeeze-excite gate and channel scaling.
#include "kernels.h"

namespace w2x {
namespace {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
// a tile pixel is four stored channels (r, g, b, 0) in the plan's precision: half4 (fp16 engines) or float4v (fp32 engines)
template <typename P> __device__ __forceinline__ P make_px(float r, float g, float b);
template <> __device__ __forceinline__ half4 make_px<half4>(float r, float g, float b) { return (half4){(_Float16)r, (_Float16)g, (_Float16)b, (_Float16)0.f}; }
template <> __device__ __forceinline__ float4v make_px<float4v>(float r, float g, float b) { return (float4v){r, g, b, 0.f}; }

// source coordinate inside the un-augmented tile for pixel (y,x) of the augmented tile (applyAugmentation)
__device__ __forceinline__ void aug_src(int k, int n, int y, int x, int& sy, int& sx) {
    switch (k) {
        default: sy = y; sx = x; break;
        case 1: sy = n - y; sx = x; break;          // flip code 0
        case 2: sy = y; sx = n - x; break;          // flip code 1
        case 3: sy = x; sx = n - y; break;          // rot90
        case 4: sy = n - y; sx = n - x; break;      // rot180
        case 5: sy = n - x; sx = y; break;          // rot270
        case 6: sy = n - x; sx = n - y; break;      // flip0 then rot90
        case 7: sy = x; sx = y; break;              // flip1 then rot90
    }
}
// source coordinate inside the network output for pixel (y,x) of the de-augmented tile (reverseAugmentation)
__device__ __forceinline__ void deaug_src(int k, int n, int y, int x, int& sy, int& sx) {
    switch (k) {
        default: sy = y; sx = x; break;
        case 1: sy = n - y; sx = x; break;
        case 2: sy = y; sx = n - x; break;
        case 3: sy = n - x; sx = y; break;          // rot270
        case 4: sy = n - y; sx = n - x; break;
        case 5: sy = x; sx = n - y; break;          // rot90
        case 6: sy = n - x; sx = n - y; break;      // rot270 then flip0
        case 7: sy = x; sx = y; break;              // rot270 then flip1
    }
}

template <typename P>
__global__ __launch_bounds__(256) void gather_kernel(const GatherParams p) {
    const int T = p.T;
    const long total = (long)p.B * T * T;
    const float inv255 = (float)(1.0 / 255.0), inv65535 = (float)(1.0 / 65535.0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / ((long)T * T));
        int rem = (int)(i - (long)b * T * T);
        int y = rem / T, x = rem - y * T;
        TileSlot sl = p.slots[b];
        P v = make_px<P>(0.f, 0.f, 0.f);
        if (sl.valid) {
            int sy, sx;
            aug_src(sl.aug, T - 1, y, x, sy, sx);
            int fy = min(max(sl.y + sy, 0), p.rows - 1);
            int fx = min(max(sl.x + sx, 0), p.cols - 1);
            if (!p.deep) {
                const uint8_t* px = p.frame + (size_t)fy * p.step + (size_t)fx * 3;
                v = make_px<P>((float)px[2] * inv255, (float)px[1] * inv255, (float)px[0] * inv255);
            } else {   // 16-bit samples (extension, README.md:88 lists it as a TODO upstream): the same conversion with 65535
                const uint16_t* px = (const uint16_t*)(p.frame + (size_t)fy * p.step) + (size_t)fx * 3;
                v = make_px<P>((float)px[2] * inv65535, (float)px[1] * inv65535, (float)px[0] * inv65535);
            }
        }
        *((P*)p.out + i) = v;
    }
}

// One output pixel: the covering tiles in ascending tile index (img2img_render.cpp:329-330), ramp weights L, T, R, B in that order on
// the clipped rect (:110-120), fp32 sums, then rint(x * 255) saturated (:342), RGB -> BGR (:343).  Returns b | g << 8 | r << 16.
__device__ __forceinline__ unsigned quantize_bgr(float r, float g, float b) {
    const unsigned B = (unsigned)min(max(__float2int_rn(b * 255.f), 0), 255), G = (unsigned)min(max(__float2int_rn(g * 255.f), 0), 255),
                   R = (unsigned)min(max(__float2int_rn(r * 255.f), 0), 255);
    return B | G << 8 | R << 16;
}
template <typename P>
__device__ __forceinline__ void compose_pixel_sums(const ComposeParams& p, const P* tiles, int X, int Y, float& r0, float& r1, float& r2) {
    const int To = p.To, n = To - 1;
    const int steps = p.tta ? 8 : 1;
    // candidate tile columns/rows: origin = idx*stride, extent To (clipped to the canvas)
    int i0 = X - To + 1; i0 = i0 <= 0 ? 0 : (i0 + p.stride_x - 1) / p.stride_x;
    int i1 = min(p.nx - 1, X / p.stride_x);
    int j0 = Y - To + 1; j0 = j0 <= 0 ? 0 : (j0 + p.stride_y - 1) / p.stride_y;
    int j1 = min(p.ny - 1, Y / p.stride_y);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    for (int ti = i0; ti <= i1; ++ti) {
        const int ox = ti * p.stride_x, lx = X - ox;
        const int rw = ox + To > p.outW ? p.outW - ox : To;
        for (int tj = j0; tj <= j1; ++tj) {
            const int oy = tj * p.stride_y, ly = Y - oy;
            const int rh = oy + To > p.outH ? p.outH - oy : To;
            const long tile = (long)ti * p.ny + tj - p.first_tile;
            const P* tp = tiles + tile * steps * (long)To * To;
            float v0, v1, v2;
            if (!p.tta) {
                const P h = tp[(long)ly * To + lx];
                v0 = (float)h[0]; v1 = (float)h[1]; v2 = (float)h[2];
            } else {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                P h;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int sy, sx;
                    deaug_src(k, n, ly, lx, sy, sx);
                    h = tp[(long)k * To * To + (long)sy * To + sx];
                    s0 += (float)h[0]; s1 += (float)h[1]; s2 += (float)h[2];
                }
                if (p.tta_bug_compat) { v0 = (float)h[0]; v1 = (float)h[1]; v2 = (float)h[2]; }
                else { v0 = s0 * 0.125f; v1 = s1 * 0.125f; v2 = s2 * 0.125f; }
            }
            if (p.ovx || p.ovy) {
                if (ox > 0 && lx < p.ovx) { float w = p.ramp_x[lx]; v0 *= w; v1 *= w; v2 *= w; }
                if (oy > 0 && ly < p.ovy) { float w = p.ramp_y[ly]; v0 *= w; v1 *= w; v2 *= w; }
                if (ox + rw < p.outW && n - lx < p.ovx) { float w = p.ramp_x[n - lx]; v0 *= w; v1 *= w; v2 *= w; }
                if (oy + rh < p.outH && n - ly < p.ovy) { float w = p.ramp_y[n - ly]; v0 *= w; v1 *= w; v2 *= w; }
            }
            acc0 += v0; acc1 += v1; acc2 += v2;
        }
    }
    r0 = acc0; r1 = acc1; r2 = acc2;
}
template <typename P>
__device__ __forceinline__ unsigned compose_pixel(const ComposeParams& p, const P* tiles, int X, int Y) {
    float a0, a1, a2;
    compose_pixel_sums<P>(p, tiles, X, Y, a0, a1, a2);
    return quantize_bgr(a0, a1, a2);
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// A thread owns four consecutive pixels of a row.  Where the four have the same covering tiles, lie inside each of them and no TTA
// is involved (all but the columns next to a tile edge), a tile contributes its four pixels as two 16-byte loads and the 12
// output bytes leave as three dwords; the per-pixel arithmetic and its order are those of compose_pixel, so the bytes are the
// same.  Everything else takes the per-pixel path.  (The output is 100 MB of u8 per 4K frame: single-byte stores and 8-byte loads
// were the kernel's bound; four pixels per thread on the per-pixel path alone lose the loads' coalescing and measured slower.)
// Launch shape (round 3): blockIdx.x / threadIdx.x pick the four-pixel column group, blockIdx.y a band of kComposeRows rows that the
// thread walks (four: more rows per thread leave too few workgroups in flight, fewer repeat the column arithmetic).  Everything that depends on the column only (covering tile columns, whether the group takes the fast path, the ramp
// weights of its four pixels) is computed once per thread, what depends on the row is the same for a whole workgroup (scalar
// registers) - the flat grid-stride loop it replaces paid a 64-bit division and six 32-bit ones per four pixels, and the kernel ran at
// 3.1 TB/s with its integer unit busier than its memory pipe.
// (measured at config 3, profiles/r3_kernels/compose_geometry.txt: rows x threads 8 x 128 0.140 ms, 4 x 128 0.119, 16 x 128 0.177, 4 x 64 0.115-0.118, 2 x 64 0.117, 1 x 64 0.125)
#ifndef W2X_COMPOSE_ROWS
#define W2X_COMPOSE_ROWS 4
#endif
#ifndef W2X_COMPOSE_THREADS
#define W2X_COMPOSE_THREADS 64
#endif
constexpr int kComposeRows = W2X_COMPOSE_ROWS, kComposeThreads = W2X_COMPOSE_THREADS;
template <typename P>
__global__ __launch_bounds__(kComposeThreads) void compose_kernel(const ComposeParams p) {
    constexpr bool kHalf = sizeof(P) == 8;                          // the four-pixel fast path reads fp16 tiles
    const int x1 = p.x1 > 0 ? p.x1 : p.outW, sw = x1 - p.x0;
    const int gw = (sw + 3) >> 2;                                   // pixel groups per row
    const int xg = blockIdx.x * kComposeThreads + threadIdx.x;
    if (xg >= gw) return;
    const P* tiles = (const P*)p.tiles;
    const int To = p.To, n = To - 1;
    const int X = p.x0 + 4 * xg;
    const int np = min(4, x1 - X);
    // tile columns covering the first and the last pixel of the group
    int a0 = X - To + 1; a0 = a0 <= 0 ? 0 : (a0 + p.stride_x - 1) / p.stride_x;
    int b0 = X + 3 - To + 1; b0 = b0 <= 0 ? 0 : (b0 + p.stride_x - 1) / p.stride_x;
    const int a1 = min(p.nx - 1, X / p.stride_x), b1 = min(p.nx - 1, (X + 3) / p.stride_x);
    const bool fast_col = kHalf && np == 4 && !p.tta && !p.deep && a0 == b0 && a1 == b1;
    const int Y0 = p.y0 + blockIdx.y * kComposeRows, Y1 = min(p.y1 > 0 ? p.y1 : p.outH, Y0 + kComposeRows);
    for (int Y = Y0; Y < Y1; ++Y) {
        uint8_t* d = p.dst + (size_t)Y * p.dst_step + (size_t)X * 3;
        unsigned px[4] = {0u, 0u, 0u, 0u};
        if (p.deep) {   // 16-bit output (extension): the same sums, rint(x * 65535) saturated, BGR
            uint16_t* d16 = (uint16_t*)(p.dst + (size_t)Y * p.dst_step) + (size_t)X * 3;
            for (int k = 0; k < np; ++k) {
                float r, g, b;
                compose_pixel_sums<P>(p, tiles, X + k, Y, r, g, b);
                d16[3 * k] = (uint16_t)min(max(__float2int_rn(b * 65535.f), 0), 65535);
                d16[3 * k + 1] = (uint16_t)min(max(__float2int_rn(g * 65535.f), 0), 65535);
                d16[3 * k + 2] = (uint16_t)min(max(__float2int_rn(r * 65535.f), 0), 65535);
            }
            continue;
        }
        const bool fast = fast_col && (((size_t)d) & 3) == 0;
        if (fast) {
            int j0 = Y - To + 1; j0 = j0 <= 0 ? 0 : (j0 + p.stride_y - 1) / p.stride_y;
            const int j1 = min(p.ny - 1, Y / p.stride_y);
            float acc[4][3] = {};
            for (int ti = a0; ti <= a1; ++ti) {
                const int ox = ti * p.stride_x, lx = X - ox;
                const int rw = ox + To > p.outW ? p.outW - ox : To;
                for (int tj = j0; tj <= j1; ++tj) {
                    const int oy = tj * p.stride_y, ly = Y - oy;
                    const int rh = oy + To > p.outH ? p.outH - oy : To;
                    const long tile = (long)ti * p.ny + tj - p.first_tile;
                    const half4* tp = (const half4*)p.tiles + tile * (long)To * To + (long)ly * To + lx;
                    half4 h[4];
                    if ((((size_t)tp) & 15) == 0) { const half8 u0 = *(const half8*)tp, u1 = *(const half8*)(tp + 2);
                        h[0] = (half4){u0[0], u0[1], u0[2], u0[3]}; h[1] = (half4){u0[4], u0[5], u0[6], u0[7]}; h[2] = (half4){u1[0], u1[1], u1[2], u1[3]}; h[3] = (half4){u1[4], u1[5], u1[6], u1[7]}; }
                    else { h[0] = tp[0]; h[1] = tp[1]; h[2] = tp[2]; h[3] = tp[3]; }
                    const bool wl = ox > 0, wt = oy > 0 && ly < p.ovy, wr = ox + rw < p.outW, wb = oy + rh < p.outH && n - ly < p.ovy;
                    const float fy_t = wt ? p.ramp_y[ly] : 1.f, fy_b = wb ? p.ramp_y[n - ly] : 1.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v0 = (float)h[k][0], v1 = (float)h[k][1], v2 = (float)h[k][2];
                        if (p.ovx || p.ovy) {
                            const int lk = lx + k;
                            if (wl && lk < p.ovx) { float w = p.ramp_x[lk]; v0 *= w; v1 *= w; v2 *= w; }
                            if (wt) { v0 *= fy_t; v1 *= fy_t; v2 *= fy_t; }
                            if (wr && n - lk < p.ovx) { float w = p.ramp_x[n - lk]; v0 *= w; v1 *= w; v2 *= w; }
                            if (wb) { v0 *= fy_b; v1 *= fy_b; v2 *= fy_b; }
                        }
                        acc[k][0] += v0; acc[k][1] += v1; acc[k][2] += v2;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) px[k] = quantize_bgr(acc[k][0], acc[k][1], acc[k][2]);
            unsigned* dw = (unsigned*)d;
            dw[0] = px[0] | px[1] << 24;
            dw[1] = px[1] >> 8 | px[2] << 16;
            dw[2] = px[2] >> 16 | px[3] << 8;
        } else {
            for (int k = 0; k < np; ++k) {
                const unsigned v = compose_pixel<P>(p, tiles, X + k, Y);
                d[3 * k] = (uint8_t)v; d[3 * k + 1] = (uint8_t)(v >> 8); d[3 * k + 2] = (uint8_t)(v >> 16);
            }
        }
    }
}

__global__ void se_kernel(const SeParams p) {
    // one block per batch item; tiny (C <= 256)
    extern __shared__ float sm[];
    float* mean = sm;            // [C]
    float* mid = sm + p.C;       // [Cmid]
    float* part = mid + p.Cmid;  // [slices][C]
    const int b = blockIdx.x;
    // add the producing GEMM's per-workgroup partial sums: the workgroups are split into `slices` contiguous ranges (one per
    // thread group), each summed in block order, then the slices are added in slice order - a fixed tree, so deterministic
    const int slices = blockDim.x / p.C > 0 ? blockDim.x / p.C : 1;
    const int per = (p.nblocks + slices - 1) / slices;
    {
        const int c = threadIdx.x % p.C, sl = threadIdx.x / p.C;
        if (sl < slices) {
            float s = 0.f;
            const int t1 = min(p.nblocks, (sl + 1) * per);
            for (int t = sl * per; t < t1; ++t) s += p.pool[((size_t)b * p.nblocks + t) * p.Cs + c];
            part[sl * p.C + c] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        float s = 0.f;
        for (int sl = 0; sl < slices; ++sl) s += part[sl * p.C + c];
        mean[c] = s * p.inv_count;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < p.Cmid; m += blockDim.x) {
        float a = p.b1[m];
        for (int c = 0; c < p.C; ++c) a += p.w1[m * p.C + c] * mean[c];
        mid[m] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.Cs; c += blockDim.x) {
        float s = 0.f;
        if (c < p.C) {
            float a = p.b2[c];
            for (int m = 0; m < p.Cmid; ++m) a += p.w2[c * p.Cmid + m] * mid[m];
            s = 1.f / (1.f + __expf(-a));
        }
        p.scale[b * p.Cs + c] = s;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(_Float16* x, const float* scale, int B, long HW, int Cs) {
    const int pc = Cs / 8;
    const long total = (long)B * HW * pc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cp = (int)(i % pc);
        long pix = i / pc;
        int b = (int)(pix / HW);
        *((half8*)x + i) = gate::gate8(*((half8*)x + i), scale + b * Cs + cp * 8);   // the arithmetic of the folded gates (kernels.h)
    }
}

__global__ __launch_bounds__(256) void scale32_kernel(float* x, const float* scale, int B, long HW, int Cs) {
    const int pc = Cs / 4;
    const long total = (long)B * HW * pc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cp = (int)(i % pc);
        const int b = (int)(i / pc / HW);
        float4v v = *((float4v*)x + i);
        const float* sc = scale + b * Cs + cp * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sc[e];
        *((float4v*)x + i) = v;
    }
}

template <typename P>
__global__ __launch_bounds__(256) void blob_to_nhwc_kernel(const float* nchw, P* out, int B, int T) {
    const long total = (long)B * T * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / ((long)T * T));
        long rem = i - (long)b * T * T;
        const float* s = nchw + (long)b * 3 * T * T + rem;
        out[i] = make_px<P>(s[0], s[(long)T * T], s[2L * T * T]);
    }
}

template <typename P>
__global__ __launch_bounds__(256) void nhwc_to_blob_kernel(const P* in, float* nchw, int B, int T) {
    const long total = (long)B * T * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / ((long)T * T));
        long rem = i - (long)b * T * T;
        const P v = in[i];
        float* d = nchw + (long)b * 3 * T * T + rem;
        d[0] = (float)v[0]; d[(long)T * T] = (float)v[1]; d[2L * T * T] = (float)v[2];
    }
}

inline unsigned grid_for(long total) { long g = (total + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace

hipError_t launch_gather(const GatherParams& p, hipStream_t s) {
    const dim3 grid(grid_for((long)p.B * p.T * p.T));
    if (p.fp32) hipLaunchKernelGGL(gather_kernel<float4v>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gather_kernel<half4>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_compose(const ComposeParams& p, hipStream_t s) {
    const int gw = (((p.x1 > 0 ? p.x1 : p.outW) - p.x0) + 3) / 4;
    if (gw <= 0 || p.outH <= 0) return hipSuccess;
    const int nrows = (p.y1 > 0 ? p.y1 : p.outH) - p.y0;
    if (nrows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((gw + kComposeThreads - 1) / kComposeThreads), (unsigned)((nrows + kComposeRows - 1) / kComposeRows));
    if (p.fp32) hipLaunchKernelGGL(compose_kernel<float4v>, grid, dim3(kComposeThreads), 0, s, p);
    else hipLaunchKernelGGL(compose_kernel<half4>, grid, dim3(kComposeThreads), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_se(const SeParams& p, hipStream_t s) {
    const int threads = 1024, slices = threads / p.C > 0 ? threads / p.C : 1;
    hipLaunchKernelGGL(se_kernel, dim3(p.B), dim3(threads), (p.C + p.Cmid + slices * p.C) * sizeof(float), s, p);
    return hipGetLastError();
}
hipError_t launch_scale(void* x, const float* scale, int B, int HW, int Cs, bool fp32, hipStream_t s) {
    if (fp32) hipLaunchKernelGGL(scale32_kernel, dim3(grid_for((long)B * HW * (Cs / 4))), dim3(256), 0, s, (float*)x, scale, B, (long)HW, Cs);
    else hipLaunchKernelGGL(scale_kernel, dim3(grid_for((long)B * HW * (Cs / 8))), dim3(256), 0, s, (_Float16*)x, scale, B, (long)HW, Cs);
    return hipGetLastError();
}
hipError_t launch_blob_to_nhwc(const float* nchw, void* out, int B, int T, bool fp32, hipStream_t s) {
    const dim3 grid(grid_for((long)B * T * T));
    if (fp32) hipLaunchKernelGGL(blob_to_nhwc_kernel<float4v>, grid, dim3(256), 0, s, nchw, (float4v*)out, B, T);
    else hipLaunchKernelGGL(blob_to_nhwc_kernel<half4>, grid, dim3(256), 0, s, nchw, (half4*)out, B, T);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_blob(const void* in, float* nchw, int B, int T, bool fp32, hipStream_t s) {
    const dim3 grid(grid_for((long)B * T * T));
    if (fp32) hipLaunchKernelGGL(nhwc_to_blob_kernel<float4v>, grid, dim3(256), 0, s, (const float4v*)in, nchw, B, T);
    else hipLaunchKernelGGL(nhwc_to_blob_kernel<half4>, grid, dim3(256), 0, s, (const half4*)in, nchw, B, T);
    return hipGetLastError();
}

}  // namespace w2x
