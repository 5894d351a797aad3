import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import __graft_entry__ as g, synth_models as sm
pkg = g.package()
path = sm.model_path('/tmp/w2x_hd', 'swin_unet/art', 4, 3)
if not os.path.exists(path): sm.export_onnx(sm.make_model('swin_unet/art', 4, seed=1237), path, 4, 256, dynamic=True)
def mk(nofuse):
    if nofuse: os.environ['W2X_NO_FUSE_HEAD'] = '1'
    else: os.environ.pop('W2X_NO_FUSE_HEAD', None)
    e = pkg.Img2Img()
    assert e.build(path, pkg.BuildConfig.fixed(4, 256)), e.last_error()
    assert e.load(path, pkg.RenderConfig(batchSize=4, height=256, width=256, scaling=4, overlap=(0.0625, 0.0625))), e.last_error()
    return e
ef, eu = mk(False), mk(True)
rng = np.random.default_rng(3)
x = rng.random((4, 3, 256, 256), dtype=np.float32)
for rep in range(3):
    yf, yu = ef.infer(x), eu.infer(x)
    d = np.abs(yf - yu)
    bad = np.argwhere(d.max(axis=1) > 0)
    print('infer rep', rep, 'max', d.max(), 'bad pixels', len(bad), bad[:3].tolist(), bad[-3:].tolist() if len(bad) else None)
frame = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
for rep in range(3):
    of, ou = ef.render(frame), eu.render(frame)
    d = np.abs(of.astype(int) - ou.astype(int)); bad = np.argwhere(d.max(axis=2) > 0)
    print('render rep', rep, 'max', d.max(), 'bad', len(bad), bad[:2].tolist(), bad[-2:].tolist() if len(bad) else None)
