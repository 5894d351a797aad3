"""Random geometries through cunet's folded launches against the launches they replace (GPU box):

    python tools/fold_stress.py [cases] [seed]

Every case draws a scale (1 / 2), a tile size (multiple of 4 in 64 ... 200), a batch (1 ... 5), TTA on / off and a frame size, builds the engine twice - as shipped, and with the
debug switches no_fuse_stem, no_fuse_up, no_conv3h_walk (stems and transposed convolutions as their own launches, image heads on the tile kernel) - and compares infer() and render()
byte for byte.  The folds run the replaced kernels' own instruction sequences, so any difference is an addressing error.  Prints one line per case; exit code 1 on a mismatch."""
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth_models as sm  # noqa: E402

pkg = importlib.import_module("waifu2x-tensorrt_amd")


def engine(path, batch, tile, scale, tta):
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(batch, tile)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=batch, height=tile, width=tile, scaling=scale, tta=tta)), eng.last_error()
    return eng


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for c in range(cases):
            scale = int(rng.choice([1, 2]))
            tile = int(rng.integers(16, 51)) * 4
            batch = int(rng.integers(1, 6))
            tta = bool(rng.integers(0, 2)) and tile <= 120
            h, w = int(rng.integers(40, 2 * tile + 60)), int(rng.integers(40, 3 * tile + 60))
            path = sm.model_path(os.path.join(tmp, f"c{c}"), "cunet/art", scale, 1)
            sm.export_onnx(sm.make_model("cunet/art", scale, seed=100 + c), path, batch=batch, tile=tile)
            frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            x = rng.random((batch, 3, tile, tile), dtype=np.float32)
            outs = []
            for off in (0, 1):
                with pkg.debug_switches(no_fuse_stem=off, no_fuse_up=off, no_conv3h_walk=off):
                    eng = engine(path, batch, tile, scale, tta)
                    outs.append((eng.infer(x), eng.render(frame)))
                eng.close()
            same = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
            bad += not same
            print(f"case {c}: scale {scale} tile {tile} batch {batch} tta {int(tta)} frame {h}x{w}: {'same bytes' if same else 'DIFFERENT'}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
