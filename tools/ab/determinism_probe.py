"""GPU box: render one cunet frame three times under a few engine switches and report how many bytes differ between renders
(debugging aid for run-to-run differences; every count must be 0)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import __graft_entry__ as g
    import synth_models as sm
    from parity_util import smooth_frame
    pkg = g.package()
    model, scale, noise = sys.argv[2], int(sys.argv[3]), 1
    path = sm.model_path("/tmp/w2x_probe", model, scale, noise)
    if not os.path.exists(path):
        sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, 256, dynamic=True)
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(4, 256)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=4, height=256, width=256, scaling=scale)), eng.last_error()
    frame = smooth_frame(1080, 1920, 7)
    outs = [eng.render(frame).copy() for _ in range(4)]
    print(sys.argv[4], [int((outs[0] != o).sum()) for o in outs[1:]], flush=True)
    eng.close()
else:
    for tag, env in (("default", {}), ("no graph", {"W2X_NO_GRAPH": "1"}), ("no split", {"W2X_NO_SPLIT": "1"}), ("no graph, no split", {"W2X_NO_GRAPH": "1", "W2X_NO_SPLIT": "1"}),
                     ("no pixgemm", {"W2X_NO_PIXGEMM": "1"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", sys.argv[1] if len(sys.argv) > 1 else "cunet/art", sys.argv[2] if len(sys.argv) > 2 else "2", tag],
                       env=dict(os.environ, **env))
