"""A few render() calls at config 3 for a rocprofv3 --kernel-trace run: which hardware queue each kernel (and the runtime's copy kernels) went to.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 tools/ab/render_trace.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import synth_models as sm
pkg = g.package()
path = sm.model_path("/tmp/w2x_render_parts", "swin_unet/art", 4, 3)
if not os.path.exists(path):
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=1237), path, 1, 256, dynamic=True)
eng = pkg.Img2Img()
assert eng.build(path, pkg.BuildConfig.fixed(4, 256)), eng.last_error()
assert eng.load(path, pkg.RenderConfig(batchSize=4, height=256, width=256, scaling=4)), eng.last_error()
frame = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
out = np.empty((4320, 7680, 3), np.uint8)
for _ in range(5):
    assert eng.render(frame, out)
