"""Diagnostic: per-phase cycle shares of the fused attention kernel (W2X_STAMPS=1)."""
import ctypes, os, sys
os.environ["W2X_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import __graft_entry__ as g, synth_models as sm
pkg = g.package()
path = sm.model_path("/tmp/w2x_stamps", "swin_unet/art", 4, 3)
if not os.path.exists(path):
    sm.export_onnx(sm.make_model("swin_unet/art", 4), path, 4, 256, dynamic=True)
eng = pkg.Img2Img()
assert eng.build(path, pkg.BuildConfig.fixed(4, 256)) and eng.load(path, pkg.RenderConfig(batchSize=4, height=256, width=256, scaling=4))
frame = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
eng.render(frame)
L = pkg.lib()
buf = (ctypes.c_ulonglong * 16)()
L.w2x_debug_attn_stamps(buf)
eng.bench_resident(2)
L.w2x_debug_attn_stamps(buf)
for title, v, names in (("barrier-staged kernel (k_swinattn.hip)", list(buf)[:8], ["gather+LN", "barrierA+stage+barrierB", "qkv products", "barrier C", "attention", "proj", "final rows", "waves"]),
                        ("C=192 register-resident kernel", list(buf)[8:], ["gather+LN+barrier", "q,k products", "v products", "S+softmax", "O+store", "barrier+proj", "barrier+final rows", "waves"])):
    tot = sum(v[:7])
    if not tot: continue
    print(title)
    for n, x in zip(names, v):
        print(f"    {n:26s} {x:16d}  {100.0 * x / tot if n != 'waves' else 0:5.1f}%  per-wave {x / max(v[7], 1):9.0f} cyc")

mb = (ctypes.c_ulonglong * 16)()
L.w2x_debug_mlp_stamps(mb)
eng.bench_resident(2)
L.w2x_debug_mlp_stamps(mb)
mn = ["load+LN", "barriers+stage", "GEMM1", "GELU", "GEMM2", "epilogue", "-", "waves"]
for ci, cname in enumerate(("mlp C=96", "mlp C=192")):
    v = list(mb)[8 * ci:8 * ci + 8]; tot = sum(v[:6])
    print(cname)
    for n, x in zip(mn, v):
        if n != "-": print(f"    {n:18s} {100.0 * x / max(tot, 1) if n != 'waves' else 0:5.1f}%  per-wave {x / max(v[7], 1):9.0f}")
