"""Ad-hoc GPU check: HIP engine vs oracle on small synthetic graphs (not a test; see tests/)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import synth_models as sm
from oracle import onnx_exec, pipeline
pkg = g.package()
work = os.path.join(ROOT, "gpurun_out", "dbg")

def check(model, scale, B, T, small=False, render=True):
    tag = f"{model} s{scale} B{B} T{T} small={small}"
    try:
        path = sm.model_path(os.path.join(work, f"{'s' if small else 'f'}{B}_{T}"), model, scale, 3)
        sm.export_onnx(sm.make_model(model, scale, seed=7, small=small), path, B, T)
        eng = pkg.Img2Img()
        if not eng.build(path, pkg.BuildConfig.fixed(B, T)): print(tag, "BUILD FAIL", eng.last_error()); return
        cfg = pkg.RenderConfig(batchSize=B, height=T, width=T, scaling=scale)
        if not eng.load(path, cfg): print(tag, "LOAD FAIL", eng.last_error()); return
        rng = np.random.default_rng(1)
        x = rng.random((B, 3, T, T), dtype=np.float32)
        x = x.astype(np.float16).astype(np.float32)
        t = time.time(); y = eng.infer(x); dt = time.time() - t
        ref = onnx_exec.Executor(path).run(x)
        d = np.abs(y - ref)
        print(f"{tag}: infer max|d|={d.max():.5f} mean|d|={d.mean():.6f} ref[min,max]=[{ref.min():.3f},{ref.max():.3f}] nan={np.isnan(y).sum()} t={dt*1e3:.1f}ms", flush=True)
        if render:
            frame = rng.integers(0, 256, (T + 37, T * 2 + 11, 3), dtype=np.uint8)
            out = eng.render(frame)
            ex = onnx_exec.Executor(path)
            refi = pipeline.render(frame, ex.run, batch=B, tile=T, scaling=scale, overlap=(0.0625, 0.0625), net_dtype=np.float16)
            di = np.abs(out.astype(int) - refi.astype(int))
            mse = float(np.mean(di.astype(np.float64) ** 2)); psnr = 99 if mse == 0 else 10 * np.log10(255 ** 2 / mse)
            print(f"{tag}: render {out.shape} max LSB diff={di.max()} psnr={psnr:.2f} ms={eng.last_render_ms:.2f}", flush=True)
    except Exception as e:
        import traceback; traceback.print_exc(); print(tag, "EXC", e, flush=True)

if __name__ == "__main__":
    check("cunet/art", 2, 1, 64)
    check("swin_unet/art", 4, 2, 64, small=True)
    check("swin_unet/art", 4, 1, 64)
    check("cunet/art", 1, 2, 64)
    check("swin_unet/art", 2, 1, 64)
