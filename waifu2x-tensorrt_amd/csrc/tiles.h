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

// One GPU's share of a single frame with EVERY TILE COMPUTED ONCE (second option: seam exchange instead of redundant columns).
// Part p of n takes the contiguous range [p * count / n, (p + 1) * count / n) of the reference's column-major tile order
// (img2img_render.cpp:43-44), so a range may begin and end inside a tile column.  The canvas is partitioned into one CELL per tile -
// cell (i, j) = x in [i * stride_x, (i + 1) * stride_x), y in [j * stride_y, (j + 1) * stride_y), the last column / row running to the canvas
// edge - and a part composes the cells of its own tiles: at most three rectangles (the tail of its first tile column, its whole columns, the
// head of its last one).  A pixel of cell (i, j) receives contributions from tiles (i-1..i, j-1..j) only (overlap < stride), added in
// ascending tile index like the reference's canvas add (img2img_render.cpp:329-330): besides its own tiles a part needs the blend bands of
// the ny + 1 tiles in front of its range - [halo_first, first_tile), computed by the preceding part(s) and copied over (the seam exchange).
struct ShardPlan {
    int first_tile = 0, tile_count = 0;   // own tiles (global indices)
    int halo_first = 0;                   // first tile whose bands are needed from the preceding parts (== first_tile: none)
    int nrect = 0;
    Rect rect[3];                         // output rectangles this part composes (x, y, w, h in canvas pixels)
};
ShardPlan shard_plan(const TileGrid& g, int outW, int outH, int tileOutW, int tileOutH, int part, int parts);

// left/top ramp: w[i] = float(double(i+1)/(ov+1)), i < ov  (img2img_load.cpp:34-45)
std::vector<float> blend_ramp(int ov);

// full mask like the reference builds it; which: 0 top, 1 right, 2 bottom, 3 left
std::vector<float> tile_weight_mask(int which, int ovx, int ovy, int size);

}  // namespace w2x
