// Host-side tile geometry: bit-exact restatement of calculateTiles / createTileWeights
// (/root/reference/src/tensorrt/img2img_render.cpp:7-66, img2img_load.cpp:29-52, 262-265).
#pragma once
#include <vector>

namespace w2x {

struct Rect { int x, y, w, h; };

struct TileGrid {
    int count = 0, nx = 0, ny = 0;
    std::vector<Rect> in, out;        // column-major: index = i*ny + j (img2img_render.cpp:43-44)
    int scaledInW = 0, scaledInH = 0; // input stride before overlap
    int inOvX = 0, inOvY = 0, outOvX = 0, outOvY = 0;
};

TileGrid calculate_tiles(int inW, int inH, int outW, int outH, int tileInW, int tileInH, int tileOutW, int tileOutH,
                         int scaling, double overlapX, double overlapY);

// left/top ramp: w[i] = float(double(i+1)/(ov+1)), i < ov  (img2img_load.cpp:34-45)
std::vector<float> blend_ramp(int ov);

// full mask like the reference builds it; which: 0 top, 1 right, 2 bottom, 3 left
std::vector<float> tile_weight_mask(int which, int ovx, int ovy, int size);

}  // namespace w2x
