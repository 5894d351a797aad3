"""Every BASELINE.json configuration that runs on a GPU, at its full size, on the kernels that ship (configs[2] is in
test_gpu_parity.py::test_headline_config_properties; configs[0] is the CPU plumbing case of tests/test_golden_e2e.py).

At full size the oracle needs minutes per frame, so each config is checked through size-independent properties - output shape,
run-to-run determinism (bit-identical), the step schedule of img2img_render.cpp:246-250 seen through the progress callback,
translation consistency (the same content under two different tiles gives the same pixels) - and against the oracle on a frame
of a few tiles of the same tile size, batch size and TTA setting.  The oracle's output for those frames is a committed fixture
(tests/golden/cfg_*.npz from tests/golden/make_config_fixtures.py: windows of the expected frame + block means of all of it;
the CPU oracle needs 13 s per 400 x 400 tile and 40 s per 640 x 640 one); W2X_LIVE_ORACLE=1 recomputes it in the test instead.  Device time per frame is printed for the record
(profiles/<round>/config_sanity.txt is produced by tools/config_sanity.py with the same settings)."""
import numpy as np
import pytest

import os

from oracle import onnx_exec, pipeline
from parity_util import BLOCK_MEAN_TOL, check_config_fixture, frame_report
from test_gpu_parity import FRAME_MAX_LSB, make_engine, oracle16, smooth_frame

pytestmark = pytest.mark.gpu


def live_oracle16(path):
    """The fp16-boundary oracle applied tile by tile to the slots that carry image data.  Batch items never interact (the ONNX batch
    axis is a plain batch), and the zero tiles the reference appends to its last batch are dropped unread
    (img2img_render.cpp:281,298-299), so skipping them changes nothing but the CPU time (14 of 16 slots at config 5)."""
    run = oracle16(path)

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


def against_the_oracle(name, tag, out, small, path, **kw):
    if os.environ.get("W2X_LIVE_ORACLE"):
        ref = pipeline.render(small, live_oracle16(path), overlap=(0.0625, 0.0625), net_dtype=np.float16, **kw)
        r = frame_report(tag, out, ref)
        assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB, r
        return
    r = check_config_fixture(name, out)
    assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB and r["max_block_mean_diff"] <= BLOCK_MEAN_TOL, r


def full_size_properties(pkg, eng, tag, hw, scale, tile, batch, tta, stride):
    rows, cols = hw
    frame = smooth_frame(rows, cols, 7)
    prog = []
    eng.setProgressCallback(lambda c, t, s: prog.append((c, t)))
    out = eng.render(frame)
    assert out.shape == (rows * scale, cols * scale, 3)
    n_tiles = pkg.calculate_tiles(cols, rows, cols * scale, rows * scale, tile, eng.output_tile_size, scale, (0.0625, 0.0625))[0]
    batches = -(-(n_tiles * (8 if tta else 1)) // batch)                  # img2img_render.cpp:246-250
    assert [c for c, _ in prog] == list(range(1, batches + 1)) and all(t == batches for _, t in prog)
    eng.setProgressCallback(None)
    again = eng.render(frame)
    assert np.array_equal(out, again), f"{tag}: render is not deterministic ({int((out != again).sum())} bytes differ)"
    assert np.array_equal(out, eng.render(frame))                          # third run = graph replay
    # translation consistency: one patch under tile (0,0) and under tile (2,1); tile origins differ by whole tile strides
    # (stride = sIn - inOverlap of calculateTiles, img2img_render.cpp:16-24,47-48), so both tiles see the same input window
    f2 = np.full_like(frame, 128)
    ps = stride - 16                                                        # the patch stays inside one tile's own region
    patch = smooth_frame(ps, ps, 11)
    dy, dx = stride, 2 * stride
    f2[16:16 + ps, 16:16 + ps] = patch
    f2[16 + dy:16 + dy + ps, 16 + dx:16 + dx + ps] = patch
    o2 = eng.render(f2)
    m = 24                                                                  # compare clear of the blend bands (inOverlap <= 40 input pixels from a tile's origin)
    a = o2[scale * (16 + m):scale * (16 + ps - m), scale * (16 + m):scale * (16 + ps - m)]
    b = o2[scale * (16 + dy + m):scale * (16 + dy + ps - m), scale * (16 + dx + m):scale * (16 + dx + ps - m)]
    assert a.shape == b.shape and np.abs(a.astype(int) - b.astype(int)).max() <= 1
    ms = eng.bench_resident(3)
    print(f"CONFIG {tag}: {cols}x{rows} -> x{scale}, {n_tiles} tiles, {batches} batches of {batch}, pass = {eng.pass_tiles} tile slots, "
          f"{ms:.2f} ms per resident frame ({rows * cols * scale * scale / 1e6 / (ms * 1e-3):.0f} MPix/s)", flush=True)


def test_config2_cunet_art_s2_n1_b4_t256_1080p(pkg, onnx_model):
    """configs[1]: cunet/art scale2 noise1 batch4 tile256 fp16, single 1920x1080 frame."""
    path = onnx_model("cunet/art", 2, 4, 256, noise=1)
    eng = make_engine(pkg, path, 4, 256, 2)
    assert eng.output_tile_size == 440
    full_size_properties(pkg, eng, "configs[1] cunet/art s2 n1 B4 T256", (1080, 1920), 2, 256, 4, False, 220 - 16)
    small = smooth_frame(300, 420, 13)                                     # 2 x 2 tiles
    out = eng.render(small)
    against_the_oracle("2", "config2[cunet/art s2 n1 B4 T256 300x420]", out, small, path, batch=4, tile=256, scaling=2, tile_out=eng.output_tile_size)
    eng.close()


def test_config4_swin_photo_s4_n3_b8_t400_tta_1080p(pkg, onnx_model):
    """configs[3]: swin_unet/photo scale4 noise3 batch8 tile400 fp16 + TTA on 1080p frames (18 tiles x 8 = 144 steps = 18 batches)."""
    path = onnx_model("swin_unet/photo", 4, 1, 400)                        # the batch axis is dynamic: trace at 1
    eng = make_engine(pkg, path, 8, 400, 4, tta=True)
    assert eng.output_tile_size == 1536
    full_size_properties(pkg, eng, "configs[3] swin_unet/photo s4 n3 B8 T400 +TTA", (1080, 1920), 4, 400, 8, True, 384 - 25)
    small = smooth_frame(120, 360, 17)                                     # one tile, 8 steps = one batch of 8 (the CPU oracle needs ~13 s per 400 x 400 tile)
    out = eng.render(small)
    against_the_oracle("4", "config4[swin_unet/photo s4 n3 B8 T400 tta 120x360]", out, small, path, batch=8, tile=400, scaling=4, tta=True, tile_out=eng.output_tile_size)
    eng.close()


def test_config5_swin_art_scan_s4_n3_b16_t640_4k(pkg, onnx_model):
    """configs[4]: swin_unet/art_scan scale4 noise3 batch16 tile640 fp16 on 3840x2160 frames (28 tiles = 2 batches of 16);
    the per-GPU hipGraph capture the config names is exercised by the third render of full_size_properties."""
    path = onnx_model("swin_unet/art_scan", 4, 1, 640)
    eng = make_engine(pkg, path, 16, 640, 4)
    assert eng.output_tile_size == 2496
    full_size_properties(pkg, eng, "configs[4] swin_unet/art_scan s4 n3 B16 T640", (2160, 3840), 4, 640, 16, False, 624 - 40)
    small = smooth_frame(200, 1100, 19)                                    # 2 x 1 tiles
    out = eng.render(small)
    against_the_oracle("5", "config5[swin_unet/art_scan s4 n3 B16 T640 200x1100]", out, small, path, batch=16, tile=640, scaling=4, tile_out=eng.output_tile_size)
    eng.close()
