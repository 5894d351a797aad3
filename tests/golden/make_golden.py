"""Generates the end-to-end golden vectors in this directory with the oracle (fp32 network, fp16 engine boundary):
    python tests/golden/make_golden.py [case ...]
Inputs are seeded; the graphs are the synthetic-weight exports of tools/synth_models.py (seed 1234+noise).
The reference itself cannot produce vectors here (TensorRT/OpenCV-CUDA absent, no ONNX weights: SURVEY.md 8c)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth_models as sm  # noqa: E402
from oracle import onnx_exec, pipeline  # noqa: E402

CASES = [
    # name, model, scale, noise, small, batch, tile, frame (rows, cols), overlap, tta
    ("cunet_s2", "cunet/art", 2, 0, False, 1, 64, (64, 80), 0.0625, False),
    ("swin_s4", "swin_unet/art", 4, 3, True, 2, 64, (50, 70), 0.0625, False),
    ("swin_s2_tta", "swin_unet/art", 2, 1, True, 4, 40, (30, 44), 0.125, True),
    # full-width graphs (96 / 192 channels): the fused attention / MLP / projection kernels of the benchmark
    ("swin_full_s4", "swin_unet/art", 4, 3, False, 2, 64, (50, 70), 0.0625, False),
    ("swin_full_s2_tta", "swin_unet/photo", 2, 1, False, 4, 64, (40, 60), 0.125, True),
]


def frame_for(name, shape):
    rng = np.random.default_rng(abs(hash(name)) % 1000 if False else sum(map(ord, name)))
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    smooth = 127 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
    img = smooth[..., None] + rng.integers(-40, 40, (*shape, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for name, model, scale, noise, small, batch, tile, shape, ov, tta in CASES:
            if len(sys.argv) > 1 and name not in sys.argv[1:]:
                continue
            path = sm.model_path(os.path.join(tmp, name), model, scale, noise)
            sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise, small=small), path, batch, tile)
            frame = frame_for(name, shape)
            ex = onnx_exec.Executor(path)
            out = pipeline.render(frame, ex.run, batch=batch, tile=tile, scaling=scale, overlap=(ov, ov), tta=tta, net_dtype=np.float16)
            np.savez_compressed(os.path.join(HERE, f"e2e_{name}.npz"), frame=frame, expected=out,
                                meta=np.array([scale, noise, int(small), batch, tile, int(tta)]), overlap=np.array([ov]), model=np.array(model))
            print(name, frame.shape, "->", out.shape)


if __name__ == "__main__":
    main()
