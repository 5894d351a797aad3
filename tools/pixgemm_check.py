"""Diagnostic: run a graph with W2X_CHECK_GENERAL=1 so every launch taken by the streaming kernels is compared with gemm_kernel."""
import os, sys
os.environ["W2X_CHECK_GENERAL"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g, synth_models as sm
pkg = g.package()
model, scale, tile = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("cunet/art", 2, 64)
path = sm.model_path("/tmp/w2x_pixchk", model, scale, 1)
sm.export_onnx(sm.make_model(model, scale, seed=7), path, 1, tile, dynamic=True)
eng = pkg.Img2Img()
eng.setMessageCallback(lambda sev, m: print(m) if ("pixgemm check" in m or "probe op" in m) else None)
assert eng.build(path, pkg.BuildConfig.fixed(1, tile)) and eng.load(path, pkg.RenderConfig(batchSize=1, height=tile, width=tile, scaling=scale)), eng.last_error()
eng.infer(np.random.default_rng(0).random((1, 3, tile, tile), dtype=np.float32))
