"""Ad-hoc: repeat render of the headline config and report where outputs differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import synth_models as sm
pkg = g.package()
work = "/tmp/dbgdet"
path = sm.model_path(work, "swin_unet/art", 4, 3)
sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=7), path, 4, 256, dynamic=True)

def smooth_frame(rows, cols, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = 120 + 70 * np.sin(xx / 11.0 + seed) * np.cos(yy / 9.0) + 30 * np.sin((xx + yy) / 23.0)
    return np.clip(img[..., None] + rng.integers(-6, 7, (rows, cols, 3)), 0, 255).astype(np.uint8)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
frames = [smooth_frame(1080, 1920, 7), smooth_frame(1080, 1920, 8), smooth_frame(300, 420, 13)]
for rnd in range(3):
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(4, 256)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=4, height=256, width=256, scaling=4)), eng.last_error()
    first = [None] * 3
    for it in range(N):
        k = it % 3
        o = eng.render(frames[k])
        if first[k] is None: first[k] = o; continue
        d = np.abs(o.astype(int) - first[k].astype(int))
        if d.max() > 0:
            nz = np.argwhere(d.max(-1) > 0)
            cells = {}
            for y, x in nz[:: max(1, len(nz) // 5000)]:
                cells[(int(x) // 896, int(y) // 896)] = cells.get((int(x) // 896, int(y) // 896), 0) + 1
            print(f"round {rnd} iter {it} frame {k}: ndiff {len(nz)} max {d.max()} y[{nz[:,0].min()},{nz[:,0].max()}] x[{nz[:,1].min()},{nz[:,1].max()}] cells {sorted(cells.items())}", flush=True)
    eng.close()
    print("round", rnd, "done", flush=True)
