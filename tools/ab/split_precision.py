import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as g
import synth_models as sm
from oracle import onnx_exec
pkg = g.package()
for model, scale, batch, tile in [("cunet/art", 2, 2, 64), ("cunet/art", 1, 1, 64), ("swin_unet/art", 4, 2, 64), ("swin_unet/photo", 2, 1, 88), ("swin_unet/art_scan", 4, 1, 64)]:
    path = sm.model_path("/tmp/w2x_split", model, scale, 1)
    if not os.path.exists(path): sm.export_onnx(sm.make_model(model, scale, seed=1235), path, 1, tile, dynamic=True)
    x = np.random.default_rng(21).random((batch, 3, tile, tile), dtype=np.float32)
    ref = onnx_exec.Executor(path).run(x).astype(np.float64)
    for prec in (pkg.Precision.FP32, pkg.Precision.TF32):
        eng = pkg.Img2Img()
        assert eng.build(path, pkg.BuildConfig.fixed(batch, tile, precision=prec)), eng.last_error()
        assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=batch, height=tile, width=tile, scaling=scale)), eng.last_error()
        y = eng.infer(x).astype(np.float64)
        d = np.abs(y - ref)
        print(f"{model} s{scale} B{batch} T{tile} {prec.name}: max {d.max():.3e} mean {d.mean():.3e} p99.9 {np.quantile(d, 0.999):.3e}", flush=True)
        eng.close()
