#include "tiles.h"

#include <cmath>

namespace w2x {

TileGrid calculate_tiles(int inW, int inH, int outW, int outH, int tileInW, int tileInH, int tileOutW, int tileOutH,
                         int scaling, double overlapX, double overlapY) {
    TileGrid g;
    // img2img_render.cpp:11-14 - the width is used for both dimensions
    const int sOutW = tileInW * scaling, sOutH = tileInW * scaling;
    // :16-19
    const int sInW = (int)std::lround((double)tileOutW / sOutW * tileInW);
    const int sInH = (int)std::lround((double)tileOutH / sOutH * tileInH);
    // :21-24
    g.inOvX = (int)std::lround(tileInW * overlapX);
    g.inOvY = (int)std::lround(tileInH * overlapY);
    // :26-29
    g.outOvX = (int)std::lround(sOutW * overlapX);
    g.outOvY = (int)std::lround(sOutH * overlapY);
    // degenerate geometry (overlap >= stride): the reference divides by zero here; report an empty grid instead
    if (sInW - g.inOvX <= 0 || sInH - g.inOvY <= 0 || sOutW <= 0) return g;
    // :31-34
    g.nx = (int)std::lround(std::ceil((double)(inW - g.inOvX) / (sInW - g.inOvX)));
    g.ny = (int)std::lround(std::ceil((double)(inH - g.inOvY) / (sInH - g.inOvY)));
    g.count = g.nx * g.ny;
    g.scaledInW = sInW; g.scaledInH = sInH;
    g.in.reserve(g.count > 0 ? g.count : 0); g.out.reserve(g.count > 0 ? g.count : 0);
    for (int i = 0; i < g.nx; ++i) {
        for (int j = 0; j < g.ny; ++j) {
            // :46-51
            g.in.push_back(Rect{-((tileInW - sInW) / 2) + i * sInW - i * g.inOvX,
                                -((tileInH - sInH) / 2) + j * sInH - j * g.inOvY, tileInW, tileInH});
            // :54-61
            const int x = i * tileOutW - i * g.outOvX;
            const int y = j * tileOutH - j * g.outOvY;
            g.out.push_back(Rect{x, y, x + tileOutW > outW ? outW - x : tileOutW, y + tileOutH > outH ? outH - y : tileOutH});
        }
    }
    return g;
}

std::vector<float> blend_ramp(int ov) {
    std::vector<float> r(ov > 0 ? ov : 0);
    const int d = ov + 1;
    for (int i = 1; i < d; ++i) r[i - 1] = (float)((double)i / d);
    return r;
}

std::vector<float> tile_weight_mask(int which, int ovx, int ovy, int size) {
    std::vector<float> m((size_t)size * size, 1.f);
    if (which == 0 || which == 2) {
        auto r = blend_ramp(ovy);
        for (int i = 0; i < ovy && i < size; ++i) {
            int row = which == 0 ? i : size - 1 - i;
            for (int x = 0; x < size; ++x) m[(size_t)row * size + x] = r[i];
        }
    } else {
        auto r = blend_ramp(ovx);
        for (int i = 0; i < ovx && i < size; ++i) {
            int col = which == 3 ? i : size - 1 - i;
            for (int y = 0; y < size; ++y) m[(size_t)y * size + col] = r[i];
        }
    }
    return m;
}

StripPlan strip_plan(const TileGrid& g, int outW, int tileOutW, int part, int parts) {
    StripPlan sp;
    if (parts <= 0 || part < 0 || part >= parts || g.nx <= 0) return sp;
    const int stride = tileOutW - g.outOvX;                   // output origin of tile column i = i * stride (img2img_render.cpp:54-55)
    // Every strip but the first also renders the k earlier tile columns whose blend band reaches into its pixels (k = 1 for every
    // blend setting of the command line, 0 without overlap), so the strips are balanced by the columns they RENDER, nx + (parts - 1) k
    // in all: strip p ends at column floor((p + 1) (nx + (parts - 1) k) / parts) - p k.  (Equal shares of own columns left the first
    // strip short and the others a column over: 20 / 30 tiles instead of 25 / 25 for nx = 9, two strips.)
    const int k = g.outOvX > 0 && stride > 0 ? (g.outOvX + stride - 1) / stride : 0;
    auto end_col = [&](int p) {
        const long t = (long)g.nx + (long)(parts - 1) * k;
        long c = (long)(p + 1) * t / parts - (long)p * k;
        if (p + 1 == parts) c = g.nx;
        return (int)std::min<long>(std::max<long>(c, 0), g.nx);
    };
    int c0 = 0;
    for (int p = 0; p < part; ++p) c0 = std::max(c0, end_col(p));
    const int c1 = std::max(c0, end_col(part));
    if (c0 >= c1) return sp;                                  // more parts than tile columns: nothing to do
    sp.x0 = c0 * stride;
    sp.x1 = c1 < g.nx ? c1 * stride : outW;
    int cc0 = c0;                                             // first tile column whose extent reaches x0
    while (cc0 > 0 && (cc0 - 1) * stride + tileOutW > sp.x0) --cc0;
    sp.first_tile = cc0 * g.ny;
    sp.tile_count = (c1 - cc0) * g.ny;
    return sp;
}

ShardPlan shard_plan(const TileGrid& g, int outW, int outH, int tileOutW, int tileOutH, int part, int parts) {
    ShardPlan sp;
    if (parts <= 0 || part < 0 || part >= parts || g.count <= 0 || g.ny <= 0) return sp;
    const int sx = tileOutW - g.outOvX, sy = tileOutH - g.outOvY;
    // a cell sees tiles (i-1..i, j-1..j) only if a tile's blend band does not reach past its neighbour: overlap < stride (every blend setting of
    // the command line: overlap <= 1/8 of the tile)
    if (sx <= 0 || sy <= 0 || g.outOvX >= sx || g.outOvY >= sy) return sp;
    const long t0 = (long)part * g.count / parts, t1 = (long)(part + 1) * g.count / parts;
    if (t0 >= t1) return sp;                                           // more parts than tiles
    sp.first_tile = (int)t0; sp.tile_count = (int)(t1 - t0);
    sp.halo_first = (int)std::max<long>(0, t0 - g.ny - 1);
    auto x_of = [&](int i) { return i >= g.nx ? outW : i * sx; };
    auto y_of = [&](int j) { return j >= g.ny ? outH : j * sy; };
    auto add = [&](int i0, int i1, int j0, int j1) {                   // cells of columns [i0, i1) x rows [j0, j1)
        if (i0 >= i1 || j0 >= j1) return;
        sp.rect[sp.nrect++] = Rect{x_of(i0), y_of(j0), x_of(i1) - x_of(i0), y_of(j1) - y_of(j0)};
    };
    const int c0 = (int)(t0 / g.ny), r0 = (int)(t0 % g.ny), c1 = (int)(t1 / g.ny), r1 = (int)(t1 % g.ny);
    if (c0 == c1) add(c0, c0 + 1, r0, r1);                             // inside one column
    else {
        int full0 = c0;
        if (r0 != 0) { add(c0, c0 + 1, r0, g.ny); full0 = c0 + 1; }    // the tail of the first column
        add(full0, c1, 0, g.ny);                                       // whole columns
        if (r1 != 0) add(c1, c1 + 1, 0, r1);                           // the head of the last column
    }
    return sp;
}

}  // namespace w2x
