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

// One GPU's share of a single frame (tile-column strips): tiles are column-major, so part p of n owns the tile
// columns [p*nx/n, (p+1)*nx/n) and composes the output columns [x0, x1) = from its first tile's origin to the next part's.
// Pixels of that range are also covered by the tail of earlier tile columns (the blend band), so those columns are
// computed redundantly: tiles [first_tile, first_tile + tile_count) are a contiguous range of the reference's tile order,
// and every output pixel receives the same contributions in the same order as in a whole-frame render (bit-identical).
struct StripPlan { int first_tile = 0, tile_count = 0, x0 = 0, x1 = 0; };
StripPlan strip_plan(const TileGrid& g, int outW, int tileOutW, int part, int parts);

// left/top ramp: w[i] = float(double(i+1)/(ov+1)), i < ov  (img2img_load.cpp:34-45)
std::vector<float> blend_ramp(int ov);

// full mask like the reference builds it; which: 0 top, 1 right, 2 bottom, 3 left
std::vector<float> tile_weight_mask(int which, int ovx, int ovy, int size);

}  // namespace w2x
