import os, sys, ctypes as C
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import synth_models as sm
pkg = g.package()
L = pkg.lib()
path = sm.model_path("/tmp/w2x_optimes", "cunet/art", 2, 1)
if not os.path.exists(path):
    sm.export_onnx(sm.make_model("cunet/art", 2, seed=1235), path, 1, 256, dynamic=True)
eng = pkg.Img2Img()
assert eng.build(path, pkg.BuildConfig.fixed(4, 256)); assert eng.load(path, pkg.RenderConfig(batchSize=4, height=256, width=256, scaling=2))
frame = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
eng.render(frame); eng.bench_resident(2)
out = (C.c_ulonglong * 6)()
L.w2x_c3p_stamps(out)            # clear
eng.bench_resident(1)
L.w2x_c3p_stamps(out)
names = ["barrier", "dma issue", "epilogue", "products+vmcnt", "chunks", "waves"]
n = out[5] or 1; ch = out[4] or 1
print({names[k]: round(out[k] / ch) for k in range(4)}, "cycles per chunk;", "chunks per wave-launch", round(ch / n, 1), "launches*waves", n)
