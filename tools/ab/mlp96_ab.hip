// Kernel-level A/B of the C = 96 fused MLP (GPU box only): the 16x16x32 schedule of round 2 (tools/ab/k_mlp96p.hip) against the
// 32x32x16 one that ships (csrc/k_mlp96q.hip) on the same random rows and weights - results against each other and against a plain
// fp32 host evaluation on the small cases, then interleaved timing rounds at the headline size (45 tiles x 240 x 240 tokens).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I waifu2x-tensorrt_amd/csrc -mllvm -amdgpu-sched-strategy=max-ilp -fno-honor-nans \
//       tools/ab/mlp96_ab.hip tools/ab/k_mlp96p.hip waifu2x-tensorrt_amd/csrc/k_mlp96q.hip -o tools/ab/mlp96_ab && tools/ab/mlp96_ab timing
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "fragorder.h"
#include "kernels.h"
namespace w2x {
hipError_t launch_mlp96p(const MlpParams& p, hipStream_t s);
hipError_t launch_mlp96q(const MlpParams& p, hipStream_t s);
}
using namespace w2x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T) + 256)); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main(int argc, char** argv) {
    const int C = 96;
    const bool timing = argc > 1;
    std::vector<long> sizes = {32L, 1000L, 64L * 100, 12L * 32 * 300 + 7};
    if (timing) sizes.push_back(2592000L);
    for (long M : sizes) for (int stats : {0, 1}) {
        if (M > 100000 && stats) continue;
        srand(C + (int)M);
        std::vector<uint16_t> x(M * C), w1(2 * C * C), w2(C * 2 * C);
        std::vector<float> b1(2 * C), b2(C);
        for (long i = 0; i < M; ++i) { const float mu = frand() * 2.f, sd = 0.3f + std::fabs(frand()) * 2.f; for (int c = 0; c < C; ++c) x[i * C + c] = f32_to_f16(mu + sd * frand()); }
        for (auto& v : w1) v = f32_to_f16(frand() * 0.2f);
        for (auto& v : w2) v = f32_to_f16(frand() * 0.1f);
        for (auto& v : b1) v = frand() * 0.5f;
        for (auto& v : b2) v = frand() * 0.1f;
        MlpParams pa; pa.M = M; pa.C = C;
        pa.x = up(x); pa.w1 = up(w1); pa.w2 = up(w2); pa.b1 = up(b1); pa.b2 = up(b2);
        MlpParams pb = pa;
        uint16_t *ya, *yb; float *sa, *sb;
        CK(hipMalloc(&ya, M * C * 2)); CK(hipMalloc(&yb, M * C * 2)); CK(hipMalloc(&sa, M * 8 + 8)); CK(hipMalloc(&sb, M * 8 + 8));
        CK(hipMemset(ya, 0xFF, M * C * 2)); CK(hipMemset(yb, 0xFF, M * C * 2));
        pa.w1_frag = up(frag_major(w1.data(), 2 * C, C)); pa.w2_frag = up(frag_w2(w2.data(), C));
        pb.w1_frag = up(frag32_major(w1.data(), 2 * C, C)); pb.w2_frag = up(frag32_w2(w2.data(), C)); pb.frag32 = true;
        pa.y = ya; pa.stats_out = stats ? sa : nullptr;
        pb.y = yb; pb.stats_out = stats ? sb : nullptr;
        CK(launch_mlp96p(pa, 0));
        CK(launch_mlp96q(pb, 0));
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> ha(M * C), hb(M * C); std::vector<float> hsa(M * 2), hsb(M * 2);
        CK(hipMemcpy(ha.data(), ya, M * C * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), yb, M * C * 2, hipMemcpyDeviceToHost));
        double md = 0, ms = 0; long bad = -1, nan = 0, ndiff = 0;
        for (long i = 0; i < M * C; ++i) { const float a = f16_to_f32(ha[i]), b = f16_to_f32(hb[i]); if (!(b == b)) ++nan; ndiff += ha[i] != hb[i]; const double d = std::fabs(a - b); if (d > md) { md = d; bad = i; } }
        if (stats) { CK(hipMemcpy(hsa.data(), sa, M * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hsb.data(), sb, M * 8, hipMemcpyDeviceToHost));
            for (long i = 0; i < M * 2; ++i) ms = std::fmax(ms, std::fabs(hsa[i] - hsb[i]) / (1e-3 + std::fabs(hsa[i]))); }
        printf("C=%d M=%ld stats=%d: 16x16 vs 32x32 max|dy|=%.5f (row %ld col %ld), %ld of %ld values differ, nan=%ld, max rel stats diff=%.2e\n", C, M, stats, md, bad / C, bad % C, ndiff, M * C, nan, ms);
        if (M <= 20000) {   // fp32 host evaluation (exact erf GELU)
            double ea = 0, eb = 0;
            std::vector<float> xn(C), hid(2 * C);
            for (long i = 0; i < M; ++i) {
                double s = 0, q = 0;
                for (int c = 0; c < C; ++c) { const float v = f16_to_f32(x[i * C + c]); s += v; q += (double)v * v; }
                const double mean = s / C, rstd = 1.0 / std::sqrt(std::max(q / C - mean * mean, 0.0) + 1e-5);
                for (int c = 0; c < C; ++c) xn[c] = f16_to_f32(f32_to_f16((float)((f16_to_f32(x[i * C + c]) - mean) * rstd)));
                for (int n = 0; n < 2 * C; ++n) {
                    double a = b1[n];
                    for (int c = 0; c < C; ++c) a += (double)xn[c] * f16_to_f32(w1[(size_t)n * C + c]);
                    hid[n] = f16_to_f32(f32_to_f16((float)(0.5 * a * (1.0 + std::erf(a / std::sqrt(2.0))))));
                }
                for (int n = 0; n < C; ++n) {
                    double a = b2[n];
                    for (int k = 0; k < 2 * C; ++k) a += (double)hid[k] * f16_to_f32(w2[(size_t)n * 2 * C + k]);
                    const double ref = a + f16_to_f32(x[i * C + n]);
                    ea = std::max(ea, std::fabs(ref - f16_to_f32(ha[i * C + n])));
                    eb = std::max(eb, std::fabs(ref - f16_to_f32(hb[i * C + n])));
                }
            }
            printf("   against the fp32 host evaluation: 16x16 max|d|=%.5f 32x32 max|d|=%.5f\n", ea, eb);
        }
        if (timing && M > 100000) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            double best[2] = {1e9, 1e9}, sum[2] = {0, 0};
            const int rounds = 8, reps = 5;
            for (int r = 0; r < rounds; ++r) for (int v = 0; v < 2; ++v) {
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < reps; ++k) CK(v ? launch_mlp96q(pb, 0) : launch_mlp96p(pa, 0));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                if (r) { best[v] = std::min(best[v], (double)ms); sum[v] += ms; }
            }
            printf("TIMING %ld rows: 16x16x32 %.4f ms (min %.4f)  32x32x16 %.4f ms (min %.4f) per launch\n", M, sum[0] / (rounds - 1), best[0], sum[1] / (rounds - 1), best[1]);
            CK(hipMemcpy(ha.data(), yb, M * C * 2, hipMemcpyDeviceToHost));
            CK(launch_mlp96q(pb, 0)); CK(hipDeviceSynchronize());
            CK(hipMemcpy(hb.data(), yb, M * C * 2, hipMemcpyDeviceToHost));
            long nd = 0; for (long i = 0; i < M * C; ++i) nd += ha[i] != hb[i];
            printf("32x32 kernel run twice: %ld elements differ\n", nd);
        }
        for (const void* d : {pa.x, pa.w1, pa.w2, (const void*)pa.b1, (const void*)pa.b2, pa.w1_frag, pa.w2_frag, pb.w1_frag, pb.w2_frag, (const void*)ya, (const void*)yb, (const void*)sa, (const void*)sb}) CK(hipFree((void*)d));
    }
    return 0;
}
