"""Oracle NETWORK outputs at the tile sizes that ship, for tests/test_gpu_parity.py::test_network_at_shipping_tile_sizes_against_fixture:

    python tests/golden/make_net_fixture.py            # -> tests/golden/net_*.npz   (CPU only; ~1 min)

Every frame-level fixture (cfg_*.npz) sees the network through u8 pixels, where one LSB is 8 fp16 ULPs of the output range.  These
fixtures keep the network's own output: the synthetic-weight graph of tools/synth_models.py (seed 1234 + noise) run by the oracle
(oracle/onnx_exec.py) on one seeded tile [1,3,T,T], once in fp32 and once in fp16-boundary mode (the oracle's model of an fp16
engine).  Kept per case: the input seed, WINDOWS of both outputs (`windows` [n,4] = y, x, h, w in output pixels; `ref32` float32 and
`ref16` float16 crops, all 3 channels) - tile corners, edges and interior - plus per-channel sums of the whole fp32 output
(`sum32`, float64).  The full [3,T',T'] output would be 28 MB at T = 400.
The reference itself cannot produce vectors here (TensorRT absent, no ONNX weights: SURVEY.md 8c); the oracle is "parity unpinned"."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth_models as sm  # noqa: E402
from oracle import onnx_exec  # noqa: E402

WIN = 128
CASES = {
    # name: model, scale, noise, tile, input seed
    "net_swin_s4_t400": ("swin_unet/photo", 4, 3, 400, 41),      # configs[3]'s tile; T = 256 (configs[1], [2]) is compared live, 2 s of oracle per tile
}


def tile_input(tile, seed):
    """the tests' input: uniform [0,1) values representable in fp16, a black and a white corner"""
    rng = np.random.default_rng(seed)
    x = rng.random((1, 3, tile, tile), dtype=np.float32).astype(np.float16).astype(np.float32)
    x[0, :, :8, :8] = 0.0; x[0, :, -8:, -8:] = 1.0
    return x


def windows_for(n):
    """corners, two edge midpoints, centre and two off-centre windows of an n x n output"""
    m = n - WIN
    pts = [(0, 0), (0, m), (m, 0), (m, m), (0, m // 2), (m // 2, 0), (m // 2, m // 2), (m // 3, (2 * m) // 3), ((2 * m) // 3, m // 3)]
    return np.array([(y, x, WIN, WIN) for y, x in pts], np.int32)


def main():
    work = os.path.join(ROOT, "gpurun_out", "net_fixture_models")
    for name, (model, scale, noise, tile, seed) in CASES.items():
        path = sm.model_path(os.path.join(work, name), model, scale, noise)
        sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, tile)
        x = tile_input(tile, seed)
        t0 = time.time()
        y32 = onnx_exec.Executor(path).run(x)[0]
        y16 = onnx_exec.Executor(path, act_dtype="float16").run(x)[0]
        wins = windows_for(y32.shape[-1])
        ref32 = np.stack([y32[:, y:y + h, x0:x0 + w] for y, x0, h, w in wins]).astype(np.float32)
        ref16 = np.stack([y16[:, y:y + h, x0:x0 + w] for y, x0, h, w in wins]).astype(np.float16)
        assert np.array_equal(ref16.astype(np.float32), np.stack([y16[:, y:y + h, x0:x0 + w] for y, x0, h, w in wins])), "fp16-boundary output is fp16"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), model=model, scale=scale, noise=noise, tile=tile, seed=seed, windows=wins,
                            ref32=ref32, ref16=ref16, sum32=y32.astype(np.float64).sum(axis=(1, 2)))
        print(f"{name}: T'={y32.shape[-1]} {len(wins)} windows, oracle {time.time() - t0:.1f} s, "
              f"fp16-boundary vs fp32 max {np.abs(y16 - y32).max() * 2048:.2f} ULP16")


if __name__ == "__main__":
    main()
