"""Oracle outputs for the BASELINE-config GPU tests (tests/test_gpu_configs.py, test_gpu_parity.py::test_headline_config_properties),
computed once on the CPU and committed as small fixtures, so that `pytest -m gpu` does not spend minutes of CPU oracle time per
run (config 4: 8 augmented 400 x 400 tiles, config 5: two 640 x 640 tiles):

    python tests/golden/make_config_fixtures.py            # -> tests/golden/cfg_*.npz

Per config: a seeded frame of one to four tiles of the config's tile size / batch size / TTA setting is rendered by the oracle
pipeline (oracle/pipeline.py) around the oracle network in fp16-boundary mode (oracle/onnx_exec.py, act_dtype=float16) on the
synthetic-weight graph of tools/synth_models.py (seed 1234 + noise).  Kept: WINDOWS of the expected u8 frame (frame corners, every
tile-seam crossing, interior points: `windows` [n,4] = y, x, h, w and `crops`) and the mean of every 32 x 32 block of the whole
frame (`block_means`, float64) - the full expected frame would be 2-10 MB per config.  The GPU tests compare the windows at
<= 1 LSB and the block means of the whole frame within BLOCK_MEAN_TOL; the bytes around the network are pinned separately, over
whole frames, by tests/test_gpu_pipeline_bytes.py.
The reference itself cannot produce vectors here (TensorRT / OpenCV-CUDA absent, no ONNX weights: SURVEY.md 8c)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_models as sm  # noqa: E402
from oracle import onnx_exec, pipeline  # noqa: E402
from parity_util import smooth_frame  # noqa: E402

WIN = 96
BLOCK = 32
CASES = {
    # name: model, scale, noise, batch, tile, tta, frame (rows, cols), frame seed
    "cfg2": ("cunet/art", 2, 1, 4, 256, False, (300, 420), 13),
    "cfg3": ("swin_unet/art", 4, 3, 4, 256, False, (300, 420), 13),
    "cfg4": ("swin_unet/photo", 4, 3, 8, 400, True, (120, 360), 17),
    "cfg5": ("swin_unet/art_scan", 4, 3, 16, 640, False, (200, 1100), 19),
}


def windows_for(out_h, out_w, out_rects, seed):
    """Deterministic window set: the four frame corners, the crossings of every tile seam, and a few seeded interior points."""
    pts = {(0, 0), (0, out_w), (out_h, 0), (out_h, out_w)}
    xs = sorted({r.x for r in out_rects if r.x > 0} | {r.x + r.w for r in out_rects if r.x + r.w < out_w})
    ys = sorted({r.y for r in out_rects if r.y > 0} | {r.y + r.h for r in out_rects if r.y + r.h < out_h})
    for x in xs or [out_w // 2]:
        for y in ys or [out_h // 2]:
            pts.add((y, x))
        pts.add((out_h // 3, x))
    for y in ys:
        pts.add((y, out_w // 3))
    rng = np.random.default_rng(seed)
    for _ in range(6):
        pts.add((int(rng.integers(0, out_h)), int(rng.integers(0, out_w))))
    wins = []
    for (y, x) in sorted(pts):
        y0 = int(np.clip(y - WIN // 2, 0, max(out_h - WIN, 0))); x0 = int(np.clip(x - WIN // 2, 0, max(out_w - WIN, 0)))
        wins.append((y0, x0, min(WIN, out_h), min(WIN, out_w)))
    return np.array(sorted(set(wins)), np.int32)


def block_means(img):
    h, w = img.shape[:2]
    hb, wb = h // BLOCK, w // BLOCK
    return img[:hb * BLOCK, :wb * BLOCK].reshape(hb, BLOCK, wb, BLOCK, 3).astype(np.float64).mean(axis=(1, 3))


def live_only(run):
    """Zero-pad slots of the last batch are never read back (img2img_render.cpp:281,298-299): skip them on the CPU."""
    def net(x):
        y = None
        for i in range(x.shape[0]):
            if not x[i].any():
                continue
            yi = run(x[i:i + 1])
            if y is None:
                y = np.zeros((x.shape[0],) + yi.shape[1:], yi.dtype)
            y[i] = yi[0]
        return y
    return net


def main(names):
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            model, scale, noise, batch, tile, tta, shape, seed = CASES[name]
            t0 = time.time()
            path = sm.model_path(os.path.join(tmp, name), model, scale, noise)
            sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, tile, dynamic=True)
            frame = smooth_frame(shape[0], shape[1], seed)
            tout = sm.output_tile_size(model, scale, tile)
            run = onnx_exec.Executor(path, act_dtype="float16").run
            out = pipeline.render(frame, live_only(run), batch=batch, tile=tile, scaling=scale, overlap=(0.0625, 0.0625), tta=tta,
                                  net_dtype=np.float16, tile_out=tout)
            _, _, rects = pipeline.calculate_tiles(shape[1], shape[0], shape[1] * scale, shape[0] * scale, (tile, tile), (tout, tout), scale, (0.0625, 0.0625))
            wins = windows_for(out.shape[0], out.shape[1], rects, seed)
            crops = np.stack([out[y:y + h, x:x + w] for y, x, h, w in wins])
            np.savez_compressed(os.path.join(HERE, f"{name}.npz".replace("cfg", "cfg_")), windows=wins, crops=crops, block_means=block_means(out),
                                meta=np.array([scale, noise, batch, tile, int(tta), shape[0], shape[1], seed]), model=np.array(model))
            print(f"{name}: {model} s{scale} B{batch} T{tile} tta{int(tta)} frame {shape} -> {out.shape}, {len(wins)} windows, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
