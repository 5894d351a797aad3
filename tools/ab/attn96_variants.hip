// Kernel-level comparison of BUILD VARIANTS of the C = 96 fused Swin attention (GPU box only; tools/ab/attn96_variants.sh builds it):
// the same source compiled NVAR times with different -D switches and -Dlaunch_swin_attn96=launch_swin_attn96_v<i>.  Every variant
// runs on the same random token maps, weights, rel-pos bias tables and shift-mask classes; variant 0 is the reference: the others
// must agree with it (and with a plain fp32 host evaluation of y = x + proj(W-MSA(LN(x))) on the small cases); then all are
// timed in interleaved rounds at the headline size (45 tiles of 240 x 240 tokens).  No oracle involved.
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
#ifndef NVAR
#define NVAR 2
#endif
hipError_t launch_swin_attn96_v0(const SwinAttnParams& p, hipStream_t s);
hipError_t launch_swin_attn96_v1(const SwinAttnParams& p, hipStream_t s);
#if NVAR > 2
hipError_t launch_swin_attn96_v2(const SwinAttnParams& p, hipStream_t s);
#endif
#if NVAR > 3
hipError_t launch_swin_attn96_v3(const SwinAttnParams& p, hipStream_t s);
#endif
#if NVAR > 4
hipError_t launch_swin_attn96_v4(const SwinAttnParams& p, hipStream_t s);
#endif
#if NVAR > 5
hipError_t launch_swin_attn96_v5(const SwinAttnParams& p, hipStream_t s);
#endif
}
typedef hipError_t (*launch_fn)(const w2x::SwinAttnParams&, hipStream_t);
static launch_fn variants[] = {w2x::launch_swin_attn96_v0, w2x::launch_swin_attn96_v1,
#if NVAR > 2
    w2x::launch_swin_attn96_v2,
#endif
#if NVAR > 3
    w2x::launch_swin_attn96_v3,
#endif
#if NVAR > 4
    w2x::launch_swin_attn96_v4,
#endif
#if NVAR > 5
    w2x::launch_swin_attn96_v5,
#endif
};
using namespace w2x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T) + 256)); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

#ifndef CW
#define CW 96      // 192: variants of csrc/k_swinattn192.hip (tools/ab/attn192_variants.sh); heads of 32, token maps of 120 x 120
#endif
#ifdef W2X_A192_STAMPS
extern "C" void w2x_sa192_stamps(unsigned long long* out);
#endif
int main(int argc, char** argv) {
    const int C = CW, NH = 6, HD = CW / 6, NTOK = 36;
    const bool timing = argc > 1;
    struct Case { int B, H, W, ry, rx, nmask; };
    std::vector<Case> cases = {{1, 12, 12, 0, 0, 1}, {2, 18, 30, 3, 3, 4}, {3, 54, 54, 3, 3, 4}, {1, 60, 60, 0, 0, 1}};
    if (timing) cases.push_back(C == 96 ? Case{45, 240, 240, 3, 3, 4} : Case{45, 120, 120, 3, 3, 4});
    if (timing && C == 192) cases.push_back(Case{45, 60, 60, 3, 3, 4});
    for (const Case& cs : cases) {
        srand(cs.B * 1000 + cs.H);
        const int B = cs.B, H = cs.H, W = cs.W, nwx = W / 6, nwin = (H / 6) * nwx;
        const long npix = (long)B * H * W;
        std::vector<uint16_t> x(npix * C), wqkv(3 * C * C), wproj(C * C);
        std::vector<float> bqkv(3 * C), bproj(C), bias32((size_t)cs.nmask * NH * 3 * 576);
        std::vector<int> maskid(nwin);
        for (long i = 0; i < npix; ++i) { const float mu = frand() * 2.f, sd = 0.3f + std::fabs(frand()); for (int c = 0; c < C; ++c) x[i * C + c] = f32_to_f16(mu + sd * frand()); }
        for (auto& v : wqkv) v = f32_to_f16(frand() * 0.15f);
        for (auto& v : wproj) v = f32_to_f16(frand() * 0.1f);
        for (auto& v : bqkv) v = frand() * 0.2f;
        for (auto& v : bproj) v = frand() * 0.1f;
        // logical table bias[m][h][q][k] (already * log2 e), classes > 0 mask some (q, k) pairs with -100 like the shifted windows
        std::vector<float> lb((size_t)cs.nmask * NH * NTOK * NTOK);
        for (int m = 0; m < cs.nmask; ++m) for (int h = 0; h < NH; ++h) for (int q = 0; q < NTOK; ++q) for (int k = 0; k < NTOK; ++k) {
            float v = frand() * 1.5f;
            if (m > 0 && (((q / (6 * (m & 1 ? 3 : 6))) != (k / (6 * (m & 1 ? 3 : 6)))) || ((m & 2) && ((q % 6) / 3 != (k % 6) / 3)))) v -= 100.f;
            lb[(((size_t)m * NH + h) * NTOK + q) * NTOK + k] = v * 1.44269504088896341f;
        }
        // the kernels' load order (lower.cpp fuse_attn): [m*NH + h][query tile 3][key tile 0 [64][4] | key tile 1 [64][4] | keys 32..35 [64]]
        for (int mh = 0; mh < cs.nmask * NH; ++mh) for (int qt = 0; qt < 3; ++qt) for (int lane = 0; lane < 64; ++lane) {
            const int fr = lane & 15, g = lane >> 4, qq = std::min(qt * 16 + fr, NTOK - 1);
            float* base = &bias32[(size_t)mh * 3 * 576 + (size_t)qt * 576];
            for (int kt = 0; kt < 2; ++kt) for (int j = 0; j < 4; ++j) base[kt * 256 + lane * 4 + j] = lb[((size_t)mh * NTOK + qq) * NTOK + kt * 16 + g * 4 + j];
            base[512 + lane] = lb[((size_t)mh * NTOK + qq) * NTOK + 32 + g];
        }
        for (int i = 0; i < nwin; ++i) maskid[i] = rand() % cs.nmask;
        SwinAttnParams p;
        p.x = up(x); p.H = H; p.W = W; p.ry = cs.ry; p.rx = cs.rx; p.B = B; p.nwin = nwin; p.C = C; p.hd = HD;
        p.wqkv = up(wqkv); p.bqkv = up(bqkv); p.scale = 1.f / std::sqrt((float)HD); p.bias32 = up(bias32); p.maskid = up(maskid);
        p.wproj = up(wproj); p.bproj = up(bproj); p.eps = 1e-5f;
        p.wqkv_frag = up(frag_major(wqkv.data(), 3 * C, C)); p.wproj_frag = up(frag_major(wproj.data(), C, C));
        uint16_t* yv[NVAR];
        std::vector<std::vector<uint16_t>> hv(NVAR, std::vector<uint16_t>(npix * C));
        for (int v = 0; v < NVAR; ++v) {
            CK(hipMalloc(&yv[v], npix * C * 2)); CK(hipMemset(yv[v], 0xFF, npix * C * 2));
            p.y = yv[v]; CK(variants[v](p, 0));
        }
        CK(hipDeviceSynchronize());
        for (int v = 0; v < NVAR; ++v) CK(hipMemcpy(hv[v].data(), yv[v], npix * C * 2, hipMemcpyDeviceToHost));
        for (int v = 1; v < NVAR; ++v) {
            double md = 0; long bad = -1, nan = 0, nd = 0;
            for (long i = 0; i < npix * C; ++i) { const float a = f16_to_f32(hv[0][i]), b = f16_to_f32(hv[v][i]); if (!(b == b)) ++nan; nd += hv[0][i] != hv[v][i]; const double d = std::fabs(a - b); if (d > md) { md = d; bad = i; } }
            printf("B=%d H=%d W=%d shift=%d nmask=%d: v0 vs v%d max|dy|=%.5f (pixel %ld ch %ld), %ld values differ, nan=%ld\n", B, H, W, cs.ry, cs.nmask, v, md, bad / C, bad % C, nd, nan);
        }
        if (npix <= 20000) {   // fp32 host evaluation
            double mr[NVAR] = {0};
            std::vector<float> xn(NTOK * C), qkv(NTOK * 3 * C), o(NTOK * C);
            std::vector<long> pix(NTOK);
            for (int b = 0; b < B; ++b) for (int wl = 0; wl < nwin; ++wl) {
                const int wy = wl / nwx, wx = wl % nwx, m = maskid[wl];
                for (int t = 0; t < NTOK; ++t) {
                    const int y = (wy * 6 + t / 6 + cs.ry) % H, xx = (wx * 6 + t % 6 + cs.rx) % W;
                    pix[t] = (long)b * H * W + (long)y * W + xx;
                    double s = 0, q = 0;
                    for (int c = 0; c < C; ++c) { const float v = f16_to_f32(x[pix[t] * C + c]); s += v; q += (double)v * v; }
                    const double mean = s / C, rstd = 1.0 / std::sqrt(std::max(q / C - mean * mean, 0.0) + 1e-5);
                    for (int c = 0; c < C; ++c) xn[t * C + c] = f16_to_f32(f32_to_f16((float)((f16_to_f32(x[pix[t] * C + c]) - mean) * rstd)));
                }
                for (int t = 0; t < NTOK; ++t) for (int n = 0; n < 3 * C; ++n) {
                    double a = bqkv[n];
                    for (int c = 0; c < C; ++c) a += (double)xn[t * C + c] * f16_to_f32(wqkv[(size_t)n * C + c]);
                    qkv[t * 3 * C + n] = (float)a;
                }
                for (int h = 0; h < NH; ++h) for (int q = 0; q < NTOK; ++q) {
                    double sc[NTOK], mx = -1e30, l = 0;
                    for (int k = 0; k < NTOK; ++k) {
                        double a = 0;
                        for (int f = 0; f < HD; ++f) a += (double)qkv[q * 3 * C + h * HD + f] * qkv[k * 3 * C + C + h * HD + f];
                        sc[k] = a * (1.0 / std::sqrt((double)HD)) * 1.44269504088896341 + lb[(((size_t)m * NH + h) * NTOK + q) * NTOK + k];
                        mx = std::max(mx, sc[k]);
                    }
                    for (int k = 0; k < NTOK; ++k) { sc[k] = std::exp2(sc[k] - mx); l += sc[k]; }
                    for (int f = 0; f < HD; ++f) { double a = 0; for (int k = 0; k < NTOK; ++k) a += sc[k] * qkv[k * 3 * C + 2 * C + h * HD + f]; o[q * C + h * HD + f] = (float)(a / l); }
                }
                for (int t = 0; t < NTOK; ++t) for (int n = 0; n < C; ++n) {
                    double a = bproj[n];
                    for (int c = 0; c < C; ++c) a += (double)o[t * C + c] * f16_to_f32(wproj[(size_t)n * C + c]);
                    const double ref = a + f16_to_f32(x[pix[t] * C + n]);
                    for (int v = 0; v < NVAR; ++v) mr[v] = std::max(mr[v], std::fabs(ref - f16_to_f32(hv[v][pix[t] * C + n])));
                }
            }
            printf("   against the fp32 host evaluation:"); for (int v = 0; v < NVAR; ++v) printf(" v%d max|d|=%.5f", v, mr[v]); printf("\n");
        }
        if (timing && B >= 45) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            double best[NVAR], sum[NVAR];
            for (int v = 0; v < NVAR; ++v) { best[v] = 1e9; sum[v] = 0; }
            const int rounds = 7, reps = 5;
            for (int r = 0; r < rounds; ++r) for (int v = 0; v < NVAR; ++v) {
                p.y = yv[v];
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < reps; ++k) CK(variants[v](p, 0));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                if (r) { best[v] = std::min(best[v], (double)ms); sum[v] += ms; }
            }
            printf("TIMING %d x %d x %d tokens, ms per launch (mean of %d rounds / min):", B, H, W, rounds - 1);
            for (int v = 0; v < NVAR; ++v) printf("  v%d %.4f / %.4f", v, sum[v] / (rounds - 1), best[v]);
            printf("\n");
            // run-to-run determinism of every variant (the timed launches above were the second .. n-th runs on the same buffers)
            std::vector<uint16_t> again(npix * C);
            for (int v = 0; v < NVAR; ++v) {
                CK(hipMemcpy(again.data(), yv[v], npix * C * 2, hipMemcpyDeviceToHost));
                long nd = 0; for (long i = 0; i < npix * C; ++i) nd += again[i] != hv[v][i];
                printf("v%d run again: %ld elements differ from its first run\n", v, nd);
            }
#ifdef W2X_A192_STAMPS
            {   // the LAST variant is the stamped build: wave cycles per phase, averaged over the waves
                unsigned long long st[8];
                w2x_sa192_stamps(st);                       // clear what the runs above left
                p.y = yv[NVAR - 1]; CK(variants[NVAR - 1](p, 0)); CK(hipDeviceSynchronize());
                w2x_sa192_stamps(st);
                static const char* names[7] = {"gather + LayerNorm + barrier", "q and k products", "v products", "S + softmax", "O + store", "barrier + proj", "barrier + residual rows"};
                double tot = 0; for (int k = 0; k < 7; ++k) tot += (double)st[k];
                printf("STAMPS (s_memtime ticks per wave, %llu waves):", st[7]);
                for (int k = 0; k < 7; ++k) printf("  %s %.0f (%.0f %%)", names[k], (double)st[k] / st[7], 100.0 * st[k] / tot);
                printf("  | total %.0f\n", tot / st[7]);
            }
#endif
        }
        for (const void* d : {p.x, p.wqkv, (const void*)p.bqkv, (const void*)p.bias32, (const void*)p.maskid, p.wproj, (const void*)p.bproj, p.wqkv_frag, p.wproj_frag}) CK(hipFree((void*)d));
        for (int v = 0; v < NVAR; ++v) CK(hipFree(yv[v]));
    }
    return 0;
}
