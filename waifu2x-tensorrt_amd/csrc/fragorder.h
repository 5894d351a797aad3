// Host-side re-ordering of fp16 weight matrices into MFMA-fragment order for the kernels that read weights straight
// from L2 (k_swinattn96.hip, k_swinattn192u.hip, k_mlp2.hip): a wave's fragment load becomes one contiguous KiB instead
// of 16 half cache lines.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace w2x {

// W [N][K] row-major -> [N/16 row tiles][K/32 k-steps][64 lanes][8]: lane = (row & 15) + 16 * g holds columns ks*32 + 8g .. +7
inline std::vector<uint16_t> frag_major(const uint16_t* w, int N, int K) {
    if (N % 16 || K % 32) throw std::runtime_error("frag_major: shape");
    std::vector<uint16_t> f((size_t)N * K);
    const int KS = K / 32;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e)
                    f[(((size_t)nt * KS + ks) * 64 + lane) * 8 + e] = w[(size_t)(nt * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8 + e];
    return f;
}

// 3x3 convolution weights W [N][K] for conv3b_kernel (k_conv3.hip) -> [N/64 blocks][K/32 k-steps][4 n-tiles][64 lanes][8]: the A rows
// of n-tile nt are a permutation of the block's channels - row r = lane & 15 holds channel 32 (nt >> 1) + 8 (r >> 2) + 4 (nt & 1) + (r & 3)
// - so the transposed product leaves a lane with two runs of 8 consecutive channels of its pixel, and the four lanes of a pixel
// write 64 contiguous bytes per store instruction
inline std::vector<uint16_t> frag_conv3b(const uint16_t* w, int N, int K) {
    if (N % 64 || K % 32) throw std::runtime_error("frag_conv3b: shape");
    std::vector<uint16_t> f((size_t)N * K);
    const int KS = K / 32;
    for (int nb = 0; nb < N / 64; ++nb)
        for (int ks = 0; ks < KS; ++ks)
            for (int nt = 0; nt < 4; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int r = lane & 15, n = nb * 64 + 32 * (nt >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
                        f[((((size_t)nb * KS + ks) * 4 + nt) * 64 + lane) * 8 + e] = w[(size_t)n * K + ks * 32 + (lane >> 4) * 8 + e];
                    }
    return f;
}

// Second MLP matrix W2 [C][2C] -> [2C/32 chunks][C/16 n-tiles][64 lanes][8]: lane (n & 15, g) holds the hidden units of the
// chunk in the order the GELU'd accumulators of the transposed first product present them (slots 0..3: 4g+j, 4..7: 16+4g+j)
inline std::vector<uint16_t> frag_w2(const uint16_t* w, int C) {
    if (C % 16) throw std::runtime_error("frag_w2: shape");
    const int H2 = 2 * C, NT = C / 16, NCH = H2 / 32;
    std::vector<uint16_t> f((size_t)C * H2);
    for (int ch = 0; ch < NCH; ++ch)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int g = lane >> 4, hid = ch * 32 + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
                    f[(((size_t)ch * NT + nt) * 64 + lane) * 8 + e] = w[(size_t)(nt * 16 + (lane & 15)) * H2 + hid];
                }
    return f;
}

// The same for v_mfma_f32_32x32x16_f16 (k_mlp96q.hip).  W [N][K] row-major -> [N/32 row tiles][K/16 k-steps][64 lanes][8]:
// lane = (row & 31) + 32 * h holds columns ks*16 + 8h .. +7
inline std::vector<uint16_t> frag32_major(const uint16_t* w, int N, int K) {
    if (N % 32 || K % 16) throw std::runtime_error("frag32_major: shape");
    std::vector<uint16_t> f((size_t)N * K);
    const int KS = K / 16;
    for (int nt = 0; nt < N / 32; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e)
                    f[(((size_t)nt * KS + ks) * 64 + lane) * 8 + e] = w[(size_t)(nt * 32 + (lane & 31)) * K + ks * 16 + (lane >> 5) * 8 + e];
    return f;
}

// Second MLP matrix W2 [C][2C] -> [2C/32 chunks][C/32 n-tiles][2 k-steps][64 lanes][8]: lane (n & 31, h) holds the hidden units of
// the chunk in the order the GELU'd 32x32 accumulator of the transposed first product presents them: registers 8s .. 8s+7 of a
// lane are rows 16s + 8(j >> 2) + 4h + (j & 3)
inline std::vector<uint16_t> frag32_w2(const uint16_t* w, int C) {
    if (C % 32) throw std::runtime_error("frag32_w2: shape");
    const int H2 = 2 * C, NT = C / 32, NCH = H2 / 32;
    std::vector<uint16_t> f((size_t)C * H2);
    for (int ch = 0; ch < NCH; ++ch)
        for (int nt = 0; nt < NT; ++nt)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int h = lane >> 5, hid = ch * 32 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
                        f[((((size_t)ch * NT + nt) * 2 + s) * 64 + lane) * 8 + e] = w[(size_t)(nt * 32 + (lane & 31)) * H2 + hid];
                    }
    return f;
}

}  // namespace w2x
