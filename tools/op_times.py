"""HIP-event time per plan op for one resident frame of any configuration (bench.py --op-times does this for config 3 only):
    python tools/op_times.py cunet/art 2 1 4 256 1080 1920 [tta] [tf32 | fp32] [switch=value ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import synth_models as sm
pkg = g.package()
model, scale, noise, batch, tile, rows, cols = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
tta = "tta" in sys.argv[8:]
for a in sys.argv[8:]:                     # debug switches (csrc/switches.h), e.g. no_conv3=1: the reference paths for A/B runs
    if "=" in a:
        k, v = a.split("=")
        assert pkg.lib().w2x_debug_set(k.encode(), int(v)), a
prec = pkg.Precision.FP32 if "fp32" in sys.argv[8:] else pkg.Precision.TF32 if "tf32" in sys.argv[8:] else pkg.Precision.FP16
path = sm.model_path("/tmp/w2x_optimes", model, scale, noise)
if not os.path.exists(path):
    sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, tile, dynamic=True)
eng = pkg.Img2Img()
assert eng.build(path, pkg.BuildConfig.fixed(batch, tile, precision=prec)), eng.last_error()
assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=batch, height=tile, width=tile, scaling=scale, tta=tta)), eng.last_error()
frame = np.random.default_rng(0).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
eng.render(frame)
eng.bench_resident(3)                      # warm-up: the second sighting of a pass captures its hipGraph
print(f"{eng.bench_resident(5):.3f} ms per resident frame, {eng.pass_tiles} tile slots per pass")
prof = eng.profile_frame()
print({k: round(v[0], 3) for k, v in prof.items() if k != "frame_ms"})
desc = pkg.describe_plan(path, eng.pass_tiles, tile, prec).splitlines()[2:]
times = list(eng.op_times())
for i, (line, t) in enumerate(zip(desc, times)):
    folded = t == 0.0 and i > 0 and " gemm " in f" {line} " and " mlp " in f" {desc[i - 1]} "      # the image head riding on the last MLP launch (engine.cpp fuse_head)
    fwd = not folded and t == 0.0 and i + 1 < len(desc) and " gemm " in f" {line} " and " gemm " in f" {desc[i + 1]} "   # the stem computed by the patch convolution's launch (fuse_stem)
    print(f"{t:8.3f} ms  {line[:170]}" + ("  (folded into the previous op's launch)" if folded else "  (computed inside the next op's launch)" if fwd else ""))
