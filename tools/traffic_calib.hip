// Calibration of the rocprofv3 HBM counters (FETCH_SIZE / WRITE_SIZE, KiB per dispatch) for the ACCESS SHAPES the kernels here use, on
// micro-kernels that move a known number of bytes (GPU box only):
//   hipcc --offload-arch=gfx950 -O2 tools/traffic_calib.hip -o tools/traffic_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -- tools/traffic_calib      (and a second pass with --pmc WRITE_SIZE)
// MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide (16 B per lane) coalesced streaming read on gfx950 and other
// widths / shapes are uncalibrated.  The kernels' shapes: flat 16-byte pieces (MLP rows), 192- and 384-byte token rows fetched by
// 16 / 32 lanes of which 12 / 24 carry data, in window (permuted) order through buffer resources (attention), LDS-DMA of fragment-major
// KiB blocks (weight staging), a weight set every workgroup re-reads (L2-resident), and the matching 16-byte stores.
// Each kernel prints its known byte count; tools/traffic_calib_summary.py divides the counters by it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }

// 1. flat streaming read, 16 B per lane (the rule's own shape)
__global__ void calib_flat_read16(const uint4v* __restrict__ src, uint4v* sink, size_t n16) {
    uint4v acc = {};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4v v = src[i]; acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3]; }
    if (acc[0] == 0x12345u) sink[0] = acc;
}
// 2 / 3. token rows of ROWB bytes fetched by LPR lanes per row (ROWB / 16 of them carry data) through a buffer resource, rows visited in
// 6 x 6 windows of a W-wide map (the attention kernels' gather); every row exactly once
template <int ROWB, int LPR>
__global__ void calib_rows_buffer(const void* src, uint4v* sink, int H, int W, int B) {
    const __amdgpu_buffer_rsrc_t X = rsrc(src, (unsigned)((size_t)B * H * W * ROWB));
    const int li = threadIdx.x % LPR, rsub = threadIdx.x / LPR, rows_per_pass = blockDim.x / LPR;
    const int nwx = W / 6, nwin = (H / 6) * nwx;
    uint4v acc = {};
    for (int iw = blockIdx.x; iw < B * nwin; iw += gridDim.x) {
        const int b = iw / nwin, wl = iw - b * nwin, wy = wl / nwx, wx = wl - wy * nwx;
        for (int t = rsub; t < 36; t += rows_per_pass) {
            const int y = wy * 6 + t / 6, x = wx * 6 + t % 6;
            const unsigned off = li < ROWB / 16 ? (unsigned)(((size_t)b * H * W + (size_t)y * W + x) * ROWB) + li * 16u : 0xFFFFFFFFu;
            const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(X, off, 0, 0);
            acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
        }
    }
    if (acc[0] == 0x12345u) sink[0] = acc;
}
// the same rows written (16-byte buffer stores, ROWB / 16 of LPR lanes)
template <int ROWB, int LPR>
__global__ void calib_rows_store(void* dst, int H, int W, int B) {
    const __amdgpu_buffer_rsrc_t Y = rsrc(dst, (unsigned)((size_t)B * H * W * ROWB));
    const int li = threadIdx.x % LPR, rsub = threadIdx.x / LPR, rows_per_pass = blockDim.x / LPR;
    const int nwx = W / 6, nwin = (H / 6) * nwx;
    for (int iw = blockIdx.x; iw < B * nwin; iw += gridDim.x) {
        const int b = iw / nwin, wl = iw - b * nwin, wy = wl / nwx, wx = wl - wy * nwx;
        for (int t = rsub; t < 36; t += rows_per_pass) {
            const int y = wy * 6 + t / 6, x = wx * 6 + t % 6;
            const unsigned off = li < ROWB / 16 ? (unsigned)(((size_t)b * H * W + (size_t)y * W + x) * ROWB) + li * 16u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b128((uint4v){(unsigned)iw, (unsigned)t, off, 7u}, Y, off, 0, 0);
        }
    }
}
// 4. LDS-DMA of contiguous KiB blocks (one wave instruction each), streaming a buffer once
__global__ void calib_lds_dma(const char* __restrict__ src, uint4v* sink, size_t nkib) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4][8][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * 4, w0 = (size_t)blockIdx.x * 4 + wv;
    int slot = 0;
    for (size_t k = w0; k < nkib; k += nw, slot = (slot + 1) & 7)
        __builtin_amdgcn_global_load_lds((const void*)(src + k * 1024 + lane * 16), (__attribute__((address_space(3))) void*)&lds[wv][slot][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (lds[wv][0][lane] == 0x5A && lds[wv][3][lane] == 0x11) sink[0] = (uint4v){1, 2, 3, 4};
}
// 5. a 288 KiB weight set that EVERY workgroup reads REP times (fragment-major KiB blocks, 16 B per lane): L2-resident after the first touch
__global__ void calib_weights_l2(const uint4v* __restrict__ w, uint4v* sink, int nkib, int rep) {
    uint4v acc = {};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = 0; r < rep; ++r)
        for (int k = wv; k < nkib; k += 4) { const uint4v v = w[(size_t)k * 64 + lane]; acc[0] ^= v[0] + r; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3]; }
    if (acc[0] == 0x12345u) sink[0] = acc;
}
// 6. flat streaming store, 16 B per lane
__global__ void calib_flat_store16(uint4v* dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint4v){(unsigned)i, 1u, 2u, 3u};
}

int main() {
    const size_t BIG = (size_t)1 << 30;                       // 1 GiB: four times the Infinity Cache, read / written once per kernel
    const int B = 45, H = 240, W = 240;                       // the C = 96 token map of the benchmark (995 MB at 192 B per row)
    const int H2 = 120;                                       // C = 192: 45 x 120 x 120 rows of 384 B (249 MB)
    void *src, *dst; uint4v* sink;
    CK(hipMalloc(&src, BIG)); CK(hipMalloc(&dst, BIG)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(src, 0x3C, BIG)); CK(hipMemset(dst, 0, BIG)); CK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    printf("KNOWN calib_flat_read16 read %zu write 0\n", BIG);
    hipLaunchKernelGGL(calib_flat_read16, dim3(grid), dim3(256), 0, 0, (const uint4v*)src, sink, BIG / 16);
    printf("KNOWN calib_rows_buffer<192,16> read %zu write 0\n", (size_t)B * H * W * 192);
    hipLaunchKernelGGL((calib_rows_buffer<192, 16>), dim3(grid), dim3(256), 0, 0, (const void*)src, sink, H, W, B);
    printf("KNOWN calib_rows_buffer<384,32> read %zu write 0\n", (size_t)B * H2 * H2 * 384);
    hipLaunchKernelGGL((calib_rows_buffer<384, 32>), dim3(grid), dim3(256), 0, 0, (const void*)src, sink, H2, H2, B);
    printf("KNOWN calib_rows_store<192,16> read 0 write %zu\n", (size_t)B * H * W * 192);
    hipLaunchKernelGGL((calib_rows_store<192, 16>), dim3(grid), dim3(256), 0, 0, dst, H, W, B);
    printf("KNOWN calib_rows_store<384,32> read 0 write %zu\n", (size_t)B * H2 * H2 * 384);
    hipLaunchKernelGGL((calib_rows_store<384, 32>), dim3(grid), dim3(256), 0, 0, dst, H2, H2, B);
    printf("KNOWN calib_lds_dma read %zu write 0\n", BIG);
    hipLaunchKernelGGL(calib_lds_dma, dim3(grid), dim3(256), 0, 0, (const char*)src, sink, BIG / 1024);
    printf("KNOWN calib_weights_l2 read %zu write 0   (288 KiB once from HBM; %d workgroups x 40 passes = %.1f GB of reads in all, L2 hits)\n", (size_t)288 * 1024, grid, grid * 40 * 288.0 * 1024 / 1e9);
    hipLaunchKernelGGL(calib_weights_l2, dim3(grid), dim3(256), 0, 0, (const uint4v*)src, sink, 288, 40);
    printf("KNOWN calib_flat_store16 read 0 write %zu\n", BIG);
    hipLaunchKernelGGL(calib_flat_store16, dim3(grid), dim3(256), 0, 0, (uint4v*)dst, BIG / 16);
    CK(hipDeviceSynchronize());
    return 0;
}
