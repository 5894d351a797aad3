// Micro-benchmark (GPU box): does fetching a 128-byte pixel as two 64-byte halves at different times (conv3_kernel's 32-channel chunks of a 64-channel map)
// cost memory throughput against fetching whole 128-byte lines?  Each workgroup copies "halo tiles" of ROWS x 64 pixels into LDS the way k_conv3.hip does.
//   mode 0: per tile, chunk 0 (bytes 0..63 of every pixel) for all rows, barrier, then chunk 1 (bytes 64..127)      [the shipped pattern, Cin = 64]
//   mode 1: per tile, whole pixels (8 lanes x 16 B per pixel)                                                         [full lines]
//   mode 2: as 0 on a map whose pixels are 64 bytes (Cin = 32): the pieces of a row are contiguous                     [the 32-channel layers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
constexpr int ROWS = 10, PIX = 64;
template <int MODE>
__global__ __launch_bounds__(256, 2) void fetch_kernel(const unsigned char* __restrict__ src, unsigned* out, int W, int H, int tiles_x, int tiles_y, int pixbytes) {
    __shared__ uint4v lds[ROWS * PIX * 4];           // 40 KB: one 32-channel chunk of the tile
    const int tid = threadIdx.x;
    const int t = blockIdx.x, img = t / (tiles_x * tiles_y), tr = t % (tiles_x * tiles_y), ty = tr / tiles_x, tx = tr % tiles_x;
    const unsigned char* base = src + ((size_t)(img * H + ty * 8) * W + tx * PIX) * pixbytes;
    const size_t rowb = (size_t)W * pixbytes;
    unsigned acc = 0;
    if (MODE == 1) {
        // 8 lanes per pixel: thread -> (pixel tid >> 3 of 32 per pass, piece tid & 7); two passes per row cover 64 pixels
        for (int half = 0; half < 2; ++half) {       // (two LDS fills of the same size as the chunked modes: rows 0..4 / 5..9)
            uint4v h[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { const int row = half * 5 + r / 2, px = (r & 1) * 32 + (tid >> 3); h[r] = *(const uint4v*)(base + row * rowb + (size_t)px * pixbytes + (tid & 7) * 16); }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROWS; ++r) lds[r * 256 + tid] = h[r];
            __syncthreads();
            acc += lds[(tid * 7) & 2047][0];
        }
    } else {
        const int nchunk = pixbytes / 64;
        for (int kc = 0; kc < nchunk; ++kc) {
            uint4v h[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) h[r] = *(const uint4v*)(base + r * rowb + (size_t)(tid >> 2) * pixbytes + kc * 64 + (tid & 3) * 16);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROWS; ++r) lds[r * 256 + tid] = h[r];
            __syncthreads();
            acc += lds[(tid * 7) & 2047][0];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const int B = 48, H = 448, W = 448;              // cunet's full-resolution maps: 48 tiles of ~442 x 442
    const int tiles_x = W / PIX, tiles_y = H / 8 - 1;
    for (int mode = 0; mode < 3; ++mode) {
        const int pixbytes = mode == 2 ? 64 : 128;
        const size_t bytes = (size_t)B * H * W * pixbytes;
        unsigned char* src; unsigned* out; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(src, 1, bytes));
        const int grid = B * tiles_x * tiles_y;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(fetch_kernel<0>, dim3(grid), dim3(256), 0, 0, src, out, W, H, tiles_x, tiles_y, pixbytes);
            else if (mode == 1) hipLaunchKernelGGL(fetch_kernel<1>, dim3(grid), dim3(256), 0, 0, src, out, W, H, tiles_x, tiles_y, pixbytes);
            else hipLaunchKernelGGL(fetch_kernel<2>, dim3(grid), dim3(256), 0, 0, src, out, W, H, tiles_x, tiles_y, pixbytes);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = ms < best ? ms : best;
        }
        const double moved = (double)grid * ROWS * PIX * pixbytes;         // bytes requested (halo rows are requested by two tiles)
        printf("mode %d (%s): %.3f ms, %.2f GB requested -> %.2f TB/s requested, map %.2f GB\n", mode,
               mode == 0 ? "two 64-byte halves of 128-byte pixels, a chunk at a time" : mode == 1 ? "whole 128-byte pixels" : "64-byte pixels (32 channels), contiguous rows",
               best, moved / 1e9, moved / best / 1e9, bytes / 1e9);
        CK(hipFree(src)); CK(hipFree(out));
    }
    return 0;
}
