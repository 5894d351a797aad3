// Timing of BUILD VARIANTS of the C = 96 fused MLP kernel (GPU box only; tools/ab/mlp96_variants.sh builds it): csrc/k_mlp96q.hip
// compiled NVAR times with different -D switches and -Dlaunch_mlp96q=launch_mlp96_v<i> -Dmlp96q_supported=mlp96_supported_v<i>.
// Variant 0 is the reference; the others report how many output values differ from it (timing experiments switch parts of the work
// off and are EXPECTED to differ), then all run in interleaved rounds at the headline size (45 tiles x 240 x 240 token rows).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "fragorder.h"
#include "kernels.h"
#ifndef NVAR
#define NVAR 2
#endif
namespace w2x {
#define DECL(i) hipError_t launch_mlp96_v##i(const MlpParams& p, hipStream_t s);
DECL(0) DECL(1)
#if NVAR > 2
DECL(2)
#endif
#if NVAR > 3
DECL(3)
#endif
#if NVAR > 4
DECL(4)
#endif
#if NVAR > 5
DECL(5)
#endif
#if NVAR > 6
DECL(6)
#endif
#if NVAR > 7
DECL(7)
#endif
}
using namespace w2x;
typedef hipError_t (*launch_fn)(const MlpParams&, hipStream_t);
static launch_fn variants[] = {launch_mlp96_v0, launch_mlp96_v1,
#if NVAR > 2
    launch_mlp96_v2,
#endif
#if NVAR > 3
    launch_mlp96_v3,
#endif
#if NVAR > 4
    launch_mlp96_v4,
#endif
#if NVAR > 5
    launch_mlp96_v5,
#endif
#if NVAR > 6
    launch_mlp96_v6,
#endif
#if NVAR > 7
    launch_mlp96_v7,
#endif
};
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T) + 256)); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

#ifndef CW
#define CW 96        // 192: variants of csrc/k_mlp2.hip (tools/ab/mlp192_variants.sh), weights in the 16x16x32 fragment order
#endif
int main(int argc, char** argv) {
    const int C = CW;
    const long M = argc > 1 ? atol(argv[1]) : (C == 96 ? 2592000L : 648000L);
    srand(7);
    std::vector<uint16_t> x(M * C), w1(2 * C * C), w2(C * 2 * C);
    std::vector<float> b1(2 * C), b2(C);
    for (long i = 0; i < M; ++i) { const float mu = frand() * 2.f, sd = 0.3f + std::fabs(frand()) * 2.f; for (int c = 0; c < C; ++c) x[i * C + c] = f32_to_f16(mu + sd * frand()); }
    for (auto& v : w1) v = f32_to_f16(frand() * 0.2f);
    for (auto& v : w2) v = f32_to_f16(frand() * 0.1f);
    for (auto& v : b1) v = frand() * 0.5f;
    for (auto& v : b2) v = frand() * 0.1f;
    MlpParams p; p.M = M; p.C = C;
    p.x = up(x); p.w1 = up(w1); p.w2 = up(w2); p.b1 = up(b1); p.b2 = up(b2);
    if (C == 96) { p.w1_frag = up(frag32_major(w1.data(), 2 * C, C)); p.w2_frag = up(frag32_w2(w2.data(), C)); p.frag32 = true; }
    else { p.w1_frag = up(frag_major(w1.data(), 2 * C, C)); p.w2_frag = up(frag_w2(w2.data(), C)); }
    // C = 192: variants built from the 32x32x16 kernel take the other fragment order - the harness keeps both copies and the launcher
    // picks by MlpParams::frag32, which variant i sets through the bit mask FRAG32_MASK (-DFRAG32_MASK=0b110: variants 1 and 2)
    MlpParams p32 = p;
    p32.w1_frag = up(frag32_major(w1.data(), 2 * C, C)); p32.w2_frag = up(frag32_w2(w2.data(), C)); p32.frag32 = true;
#ifndef FRAG32_MASK
#define FRAG32_MASK 0
#endif
    auto params_of = [&](int v) -> MlpParams& { return (C == 192 && ((FRAG32_MASK >> v) & 1)) ? p32 : p; };
    uint16_t* yv[NVAR];
    std::vector<uint16_t> h0(M * C), hv(M * C);
    for (int v = 0; v < NVAR; ++v) {
        CK(hipMalloc(&yv[v], M * C * 2)); CK(hipMemset(yv[v], 0xFF, M * C * 2));
        params_of(v).y = yv[v]; CK(variants[v](params_of(v), 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(v ? hv.data() : h0.data(), yv[v], M * C * 2, hipMemcpyDeviceToHost));
        if (v) {
            double md = 0; long nd = 0, nan = 0;
            for (long i = 0; i < M * C; ++i) { const float a = f16_to_f32(h0[i]), b = f16_to_f32(hv[i]); if (!(b == b)) { ++nan; continue; } nd += h0[i] != hv[i]; md = std::max(md, (double)std::fabs(a - b)); }
            printf("v%d vs v0: max|dy|=%.5f, %ld of %ld values differ, nan=%ld\n", v, md, nd, M * C, nan);
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best[NVAR], sum[NVAR];
    for (int v = 0; v < NVAR; ++v) { best[v] = 1e9; sum[v] = 0; }
    const int rounds = 7, reps = getenv("W2X_AB_REPS") ? atoi(getenv("W2X_AB_REPS")) : 5;      // (W2X_AB_REPS=600: half a second per variant and round, long enough for tools/power_trace.py to see its power)
    for (int r = 0; r < rounds; ++r) for (int v = 0; v < NVAR; ++v) {
        params_of(v).y = yv[v];
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; ++k) CK(variants[v](params_of(v), 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        if (r) { best[v] = std::min(best[v], (double)ms); sum[v] += ms; }
    }
    printf("TIMING %ld rows, ms per launch (mean of %d rounds / min):", M, rounds - 1);
    for (int v = 0; v < NVAR; ++v) printf("  v%d %.4f / %.4f", v, sum[v] / (rounds - 1), best[v]);
    printf("\n");
    return 0;
}
