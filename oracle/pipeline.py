"""ORACLE (test infrastructure, not product code): numpy restatement of the reference's
tile pipeline around the network - trt::Img2Img::render / infer.

Every function cites the reference lines it follows (paths relative to /root/reference/).
The reference runs these steps through OpenCV-CUDA calls (cv::cuda::copyMakeBorder, flip,
rotate, multiply, add, convertTo, cvtColor); OpenCV is a third-party dependency that is not
vendored and not installed, so their published semantics are restated:
  * copyMakeBorder(BORDER_REPLICATE)  = clamp-to-edge indexing
  * flip(code 0) reverses rows, flip(code 1) reverses columns
  * rotate(90 | 180 | 270 with the shifts used) = numpy.rot90(x, 1 | 2 | 3)  (counter-clockwise)
  * convertTo(CV_32F, a): f32 = float(u8) * float(a);  convertTo(CV_8U, a): saturate(rint(f32*a))
    with round-half-to-even
  * multiply / add on CV_32F: plain IEEE fp32 elementwise ops.
PARITY PINNING: the reference has no tests, golden vectors or fixtures;
the tile-grid / ramp known-answer vectors in tests/golden/ were derived by evaluating the
reference's formulas (img2img_render.cpp:7-66, img2img_load.cpp:29-52) by hand in SURVEY.md
section 8c and are checked in tests/test_oracle_pipeline.py.  The network itself is
"parity unpinned" (TensorRT is closed and absent).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


def c_lround(v: float) -> int:
    """C lround(): round half away from zero (used at img2img_render.cpp:17-34)."""
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


@dataclass
class Rect:
    x: int
    y: int
    w: int
    h: int

    def astuple(self):
        return (self.x, self.y, self.w, self.h)


def calculate_tiles(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, overlap):
    """calculateTiles, img2img_render.cpp:7-66.  tile_in/tile_out are (w,h); overlap is (x,y).
    Returns (tile_count, [input rects], [output rects]) in column-major tile order (:43-44)."""
    tiw, tih = tile_in
    tow, toh = tile_out
    # :11-14  (width used for both dims - quirk Q5)
    sow = tiw * scaling
    soh = tiw * scaling
    # :16-19
    siw = c_lround(float(tow) / sow * tiw)
    sih = c_lround(float(toh) / soh * tih)
    # :21-24
    iox = c_lround(tiw * overlap[0])
    ioy = c_lround(tih * overlap[1])
    # :26-29
    oox = c_lround(sow * overlap[0])
    ooy = c_lround(soh * overlap[1])
    # :31-34
    nx = c_lround(math.ceil(float(in_w - iox) / (siw - iox)))
    ny = c_lround(math.ceil(float(in_h - ioy) / (sih - ioy)))
    ins, outs = [], []
    for i in range(nx):
        for j in range(ny):
            # :46-51 ; C++ int division truncates toward zero
            bx = int((tiw - siw) / 2)
            by = int((tih - sih) / 2)
            ins.append(Rect(-bx + i * siw - i * iox, -by + j * sih - j * ioy, tiw, tih))
            # :54-61
            x = i * tow - i * oox
            y = j * toh - j * ooy
            outs.append(Rect(x, y,
                             out_w - x if x + tow > out_w else tow,
                             out_h - y if y + toh > out_h else toh))
    return nx * ny, ins, outs


def pad_roi(img: np.ndarray, r: Rect) -> np.ndarray:
    """padRoi, img2img_render.cpp:68-105: crop with BORDER_REPLICATE for the out-of-frame part."""
    H, W = img.shape[:2]
    ys = np.clip(np.arange(r.y, r.y + r.h), 0, H - 1)
    xs = np.clip(np.arange(r.x, r.x + r.w), 0, W - 1)
    return img[ys][:, xs]


def create_tile_weights(overlap_xy, size_wh):
    """createTileWeights, img2img_load.cpp:29-52.  Returns [top, right, bottom, left] fp32 HxW
    (the reference keeps 3 identical channels)."""
    ovx, ovy = overlap_xy
    w, h = size_wh
    top = np.ones((h, w), np.float32)
    left = np.ones((h, w), np.float32)
    height = ovy + 1
    for i in range(1, height):
        top[i - 1, :] = np.float32(float(i) / height)      # :36-37 double -> Scalar -> float
    width = ovx + 1
    for i in range(1, width):
        left[:, i - 1] = np.float32(float(i) / width)       # :43-44
    bottom = top[::-1, :].copy()                              # :48 flip code 0
    right = left[:, ::-1].copy()                              # :51 flip code 1
    return [top, right, bottom, left]


def apply_weights(tile: np.ndarray, rect: Rect, out_w, out_h, weights) -> np.ndarray:
    """applyWeights, img2img_render.cpp:107-121: in-place fp32 multiplies, order L,T,R,B."""
    t = tile.astype(np.float32, copy=True)
    if rect.x > 0:
        t *= weights[3][..., None]
    if rect.y > 0:
        t *= weights[0][..., None]
    if rect.x + rect.w < out_w:
        t *= weights[1][..., None]
    if rect.y + rect.h < out_h:
        t *= weights[2][..., None]
    return t


def apply_augmentation(x: np.ndarray, k: int) -> np.ndarray:
    """applyAugmentation, img2img_render.cpp:134-177 on HxWxC arrays (enum :123-132)."""
    if k == 0:
        return x
    if k == 1:
        return x[::-1]                       # flip code 0
    if k == 2:
        return x[:, ::-1]                    # flip code 1
    if k == 3:
        return np.rot90(x, 1)
    if k == 4:
        return np.rot90(x, 2)
    if k == 5:
        return np.rot90(x, 3)
    if k == 6:
        return np.rot90(x[::-1], 1)
    if k == 7:
        return np.rot90(x[:, ::-1], 1)
    raise ValueError(k)


def reverse_augmentation(x: np.ndarray, k: int) -> np.ndarray:
    """reverseAugmentation, img2img_render.cpp:179-222."""
    if k == 0:
        return x
    if k == 1:
        return x[::-1]
    if k == 2:
        return x[:, ::-1]
    if k == 3:
        return np.rot90(x, 3)
    if k == 4:
        return np.rot90(x, 2)
    if k == 5:
        return np.rot90(x, 1)
    if k == 6:
        return np.rot90(x, 3)[::-1]
    if k == 7:
        return np.rot90(x, 3)[:, ::-1]
    raise ValueError(k)


def blob_from_tiles(tiles_u8_rgb) -> np.ndarray:
    """blobFromImages, img2img_infer.cpp:5-21: HWC u8 -> NCHW f32, f32 = u8 * float(1/255).
    (uint16 tiles - the 16-bit extension, no reference counterpart - use float(1/65535) the same way.)"""
    a = np.stack(tiles_u8_rgb)
    scale = np.float32(1.0 / 255.0) if a.dtype == np.uint8 else np.float32(1.0 / 65535.0)
    a = a.astype(np.float32) * scale
    return np.ascontiguousarray(a.transpose(0, 3, 1, 2))


def to_u8(canvas_f32: np.ndarray) -> np.ndarray:
    """convertTo(CV_8UC3, 255.0), img2img_render.cpp:342: saturate_cast<uchar>(rint(v*255))."""
    v = np.rint(canvas_f32.astype(np.float32) * np.float32(255.0))
    return np.clip(v, 0, 255).astype(np.uint8)


def to_u16(canvas_f32: np.ndarray) -> np.ndarray:
    """The 16-bit extension's counterpart of to_u8: convertTo(CV_16UC3, 65535.0) = saturate_cast<ushort>(rint(v*65535))."""
    v = np.rint(canvas_f32.astype(np.float32) * np.float32(65535.0))
    return np.clip(v, 0, 65535).astype(np.uint16)


def render(frame_bgr: np.ndarray, net, *, batch, tile, scaling, overlap, tta=False,
           tta_bug_compat=False, net_dtype=None, progress=None, tile_out=None) -> np.ndarray:
    """trt::Img2Img::render, img2img_render.cpp:224-352.

    frame_bgr: HxWx3 u8 (ffmpeg bgr24, capture.cpp:99).  net(x[B,3,T,T] f32) -> [B,3,T',T'] f32.
    overlap: (x,y) blend fractions (RenderConfig::overlap, config.h:41).
    net_dtype: if np.float16, the network input/output are rounded to fp16 like an fp16 engine
    would (values crossing the engine boundary stay f32 in the reference: img2img_load.cpp:230).
    tta_bug_compat reproduces quirk Q1 (:313-316: the mean is computed, then the last de-augmented
    output is blended instead); default is the intended mean.
    """
    H, W = frame_bgr.shape[:2]
    rgb = frame_bgr[..., ::-1]                                        # :227
    out_h, out_w = H * scaling, W * scaling
    canvas = np.zeros((out_h, out_w, 3), np.float32)                  # :228-229
    # T' is discovered from the engine's output tensor shape (img2img_load.cpp:203); callers that know it pass tile_out,
    # otherwise one zero batch is pushed through the network to read it off
    tout = int(tile_out) if tile_out else net(np.zeros((batch, 3, tile, tile), np.float32)).shape[-1]
    count, in_rects, out_rects = calculate_tiles(W, H, out_w, out_h, (tile, tile), (tout, tout),
                                                 scaling, overlap)
    overlapping = overlap[0] != 0 or overlap[1] != 0                  # :244
    weights = None
    if overlapping:                                                   # img2img_load.cpp:262-269
        ov = (c_lround(tile * scaling * overlap[0]), c_lround(tile * scaling * overlap[1]))
        weights = create_tile_weights(ov, (tout, tout))
    steps_per_tile = 8 if tta else 1                                  # :246-248
    batch_count = c_lround(math.ceil(float(count * steps_per_tile) / batch))
    step_count = batch_count * batch
    queue, tiles = [], []
    tta_acc = None
    for step in range(step_count):                                    # :260
        ti, aug, bi = step // steps_per_tile, step % steps_per_tile, step % batch
        queue.append((ti, aug))
        if ti < count:
            t = pad_roi(rgb, in_rects[ti])                            # :271
            if tta and aug != 0:
                t = apply_augmentation(t, aug)                        # :274
            tiles.append(np.ascontiguousarray(t))
        else:
            tiles.append(np.zeros((tile, tile, 3), frame_bgr.dtype))  # :281 zero pad slot
        if bi != batch - 1:
            continue
        x = blob_from_tiles(tiles)                                    # infer(), img2img_infer.cpp:73
        if net_dtype is not None:
            x = x.astype(net_dtype).astype(np.float32)
        y = net(x)                                                    # :80 enqueueV3
        if net_dtype is not None:
            y = y.astype(net_dtype).astype(np.float32)
        outs = [np.ascontiguousarray(y[b].transpose(1, 2, 0)) for b in range(batch)]  # imagesFromBlob
        for b in range(batch):                                        # :296
            ti, aug = queue[0]
            if ti == count:
                break
            queue.pop(0)
            o = outs[b]
            if tta:                                                   # :305-318
                if aug == 0:
                    tta_acc = np.zeros_like(o)
                    tta_acc = tta_acc + o
                else:
                    d = np.ascontiguousarray(reverse_augmentation(o, aug))
                    tta_acc = tta_acc + d
                    if aug == 7:
                        tta_acc = tta_acc * np.float32(1.0 / 8)
                        o = d if tta_bug_compat else tta_acc
                if aug != 7:
                    continue
            r = out_rects[ti]
            if overlapping:
                o = apply_weights(o, r, out_w, out_h, weights)        # :326
            canvas[r.y:r.y + r.h, r.x:r.x + r.w] += o[:r.h, :r.w]     # :329-330
        tiles = []
        if progress:
            progress(step // batch + 1, batch_count)
    out = to_u8(canvas) if frame_bgr.dtype == np.uint8 else to_u16(canvas)   # :342
    return np.ascontiguousarray(out[..., ::-1])                       # :343 RGB2BGR
