// Kernel-level A/B check (GPU box only):  hipcc --offload-arch=gfx950 -O2 -std=c++17 -I waifu2x-tensorrt_amd/csrc tools/ab/mlp_ab.hip \
//     tools/ab/k_mlp_staged.hip waifu2x-tensorrt_amd/csrc/k_mlp2.hip -o /tmp/mlp_ab && /tmp/mlp_ab
// Runs the LDS-staged (tools/ab/k_mlp_staged.hip, round 1) and the wave-private (k_mlp2.hip, shipped) fused MLP kernels on the same random rows/weights
// and prints the largest difference, with and without the LayerNorm-statistics output.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "fragorder.h"
#include "kernels.h"
using namespace w2x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T) + 256)); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
int main() {
    for (int C : {96, 192}) for (long M : {64L * 100, 288L, 1152L, 1000L, C == 96 ? 2592000L : 648000L}) for (int stats : {0, 1}) {
        if (M > 100000 && stats) continue;
        srand(C + (int)M);
        std::vector<uint16_t> x(M * C), w1(2 * C * C), w2(C * 2 * C);
        std::vector<float> b1(2 * C), b2(C);
        for (auto& v : x) v = f32_to_f16(frand() * 2.f);
        for (auto& v : w1) v = f32_to_f16(frand() * 0.1f);
        for (auto& v : w2) v = f32_to_f16(frand() * 0.1f);
        for (auto& v : b1) v = frand() * 0.1f;
        for (auto& v : b2) v = frand() * 0.1f;
        MlpParams p; p.M = M; p.C = C;
        p.x = up(x); p.w1 = up(w1); p.w2 = up(w2); p.b1 = up(b1); p.b2 = up(b2);
        uint16_t *ya, *yb; float *sa, *sb;
        CK(hipMalloc(&ya, M * C * 2)); CK(hipMalloc(&yb, M * C * 2)); CK(hipMalloc(&sa, M * 8)); CK(hipMalloc(&sb, M * 8));
        p.y = ya; p.stats_out = stats ? sa : nullptr;
        CK(launch_mlp(p, 0));                       // no fragment copies -> k_mlp.hip
        p.w1_frag = up(frag_major(w1.data(), 2 * C, C)); p.w2_frag = up(frag_w2(w2.data(), C));
        p.y = yb; p.stats_out = stats ? sb : nullptr;
        CK(launch_mlp(p, 0));                       // k_mlp2.hip
        CK(hipDeviceSynchronize());
        if (M > 100000) {   // determinism of the wave-private kernel at full size: run it again into the first buffer
            p.y = ya; CK(launch_mlp(p, 0)); CK(hipDeviceSynchronize());
            std::vector<uint16_t> h1(M * C), h2(M * C);
            CK(hipMemcpy(h1.data(), ya, M * C * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), yb, M * C * 2, hipMemcpyDeviceToHost));
            long nd = 0, first = -1; for (long i = 0; i < M * C; ++i) if (h1[i] != h2[i]) { if (first < 0) first = i; ++nd; }
            printf("C=%d M=%ld: wave-private kernel run twice: %ld elements differ (first at row %ld col %ld)\n", C, M, nd, first / C, first % C);
            { const long BMr = C == 96 ? 256 : 128; long hist[12] = {0};
              for (long i = 0; i < M * C; ++i) if (h1[i] != h2[i]) { long wg = i / C / BMr; int b = wg < 256 ? 0 : wg < 512 ? 1 : wg < 768 ? 2 : wg < 1024 ? 3 : wg < 2048 ? 4 : 5; hist[b]++; hist[6 + (int)((i / C % BMr) / (BMr / 4))]++; }
              printf("   differing elements by workgroup index [0,256) [256,512) [512,768) [768,1024) [1024,2048) rest: %ld %ld %ld %ld %ld %ld; by wave in workgroup: %ld %ld %ld %ld\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8], hist[9]); }
            p.y = ya; const void* f1 = p.w1_frag; p.w1_frag = nullptr; CK(launch_mlp(p, 0)); CK(hipDeviceSynchronize()); p.w1_frag = f1;   // restore the staged result for the A/B below
        }
        std::vector<uint16_t> ha(M * C), hb(M * C); std::vector<float> hsa(M * 2), hsb(M * 2);
        CK(hipMemcpy(ha.data(), ya, M * C * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), yb, M * C * 2, hipMemcpyDeviceToHost));
        double md = 0, ms = 0; long bad_row = -1;
        for (long i = 0; i < M * C; ++i) { double d = std::fabs(f16_to_f32(ha[i]) - f16_to_f32(hb[i])); if (d > md) { md = d; bad_row = i / C; } }
        if (stats) { CK(hipMemcpy(hsa.data(), sa, M * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hsb.data(), sb, M * 8, hipMemcpyDeviceToHost));
            for (long i = 0; i < M * 2; ++i) ms = std::fmax(ms, std::fabs(hsa[i] - hsb[i]) / (1e-3 + std::fabs(hsa[i]))); }
        printf("C=%d M=%ld stats=%d: max|dy|=%.5f (row %ld) max rel stats diff=%.2e\n", C, M, stats, md, bad_row, ms);
        if (md > 0.01 && stats == 0) {
            const int RW = C == 96 ? 64 : 32;
            long byt[4] = {0, 0, 0, 0}, bynt[12] = {0}, byrow16[16] = {0}, total = 0;
            for (long i = 0; i < M * C; ++i) { double d = std::fabs(f16_to_f32(ha[i]) - f16_to_f32(hb[i])); if (d > 0.01) { long r = i / C, c = i % C; byt[(r % RW) / 16]++; bynt[c / 16]++; byrow16[r % 16]++; total++; } }
            printf("   bad=%ld of %ld; by token tile:", total, M * C); for (int t = 0; t < RW / 16; ++t) printf(" %ld", byt[t]);
            printf("; by n-tile:"); for (int t = 0; t < C / 16; ++t) printf(" %ld", bynt[t]);
            printf("; by row%%16:"); for (int t = 0; t < 16; ++t) printf(" %ld", byrow16[t]); printf("\n");
        }
    }
    return 0;
}
