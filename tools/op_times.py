"""Per-op HIP-event times of one resident frame for any model/config:  op_times.py MODEL SCALE NOISE BATCH TILE [H W]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g, synth_models as sm
pkg = g.package()
model, scale, noise, batch, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
H, W = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (1080, 1920)
path = sm.model_path("/tmp/w2x_optimes", model, scale, noise)
if not os.path.exists(path):
    sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, tile, dynamic=True)
eng = pkg.Img2Img()
assert eng.build(path, pkg.BuildConfig.fixed(batch, tile)) and eng.load(path, pkg.RenderConfig(batchSize=batch, height=tile, width=tile, scaling=scale)), eng.last_error()
frame = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
eng.render(frame)
ms = eng.bench_resident(5)
prof = eng.profile_frame()
desc = pkg.describe_plan(path, eng.pass_tiles, tile).splitlines()[2:]
for line, t in zip(desc, eng.op_times()):
    print(f"{t:8.3f} ms  {line[:140]}")
print(f"frame {ms:.2f} ms resident; families {dict((k, round(v[0], 3)) for k, v in prof.items() if k != 'frame_ms')}; op times are for the LAST network pass of the frame")
