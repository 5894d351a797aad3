"""GPU sanity of the BASELINE.json configs that bench.py does not time (2, 4, 5): build, render a full-size synthetic frame,
check determinism and report device ms/frame; for the large-tile graphs also compare one small frame with the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g
import synth_models as sm
from oracle import onnx_exec, pipeline
pkg = g.package()
work = "/tmp/w2x_cfg"

def _memory_watchdog(limit_gb=48.0):
    """The GPU box must never be driven out of host memory: leave at once if this process grows beyond the limit."""
    import threading, psutil
    me = psutil.Process()
    def loop():
        while True:
            if me.memory_info().rss > limit_gb * 2 ** 30:
                print(f"memory watchdog: RSS above {limit_gb} GB, exiting", flush=True)
                os._exit(3)
            time.sleep(0.25)
    threading.Thread(target=loop, daemon=True).start()
_memory_watchdog()

def frame(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 120 + 70 * np.sin(xx / 11.0 + seed) * np.cos(yy / 9.0) + 30 * np.sin((xx + yy) / 23.0)
    return np.clip(img[..., None] + rng.integers(-6, 7, (h, w, 3)), 0, 255).astype(np.uint8)

def run(tag, model, scale, noise, batch, tile, hw, tta=False, oracle_hw=None):
    path = sm.model_path(os.path.join(work, tag), model, scale, noise)
    if not os.path.exists(path):
        sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, tile, dynamic=True)   # batch axis is dynamic: trace small
    eng = pkg.Img2Img()
    t0 = time.time()
    assert eng.build(path, pkg.BuildConfig.fixed(batch, tile)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=batch, height=tile, width=tile, scaling=scale, tta=tta)), eng.last_error()
    t1 = time.time()
    f = frame(hw[0], hw[1], 3)
    a = eng.render(f); b = eng.render(f)
    ms = eng.bench_resident(3)
    n = pkg.calculate_tiles(hw[1], hw[0], hw[1] * scale, hw[0] * scale, tile, eng.output_tile_size, scale, (0.0625, 0.0625))[0]
    msg = f"{tag}: {model} s{scale} B{batch} T{tile} tta={tta} {hw[1]}x{hw[0]}: {n} tiles, pass={eng.pass_tiles} slots, build+load {t1 - t0:.1f}s, " \
          f"{ms:.2f} ms/frame resident ({hw[0] * hw[1] * scale * scale / 1e6 / (ms * 1e-3):.0f} MPix/s), deterministic={np.array_equal(a, b)}, " \
          f"out range [{a.min()},{a.max()}]"
    if oracle_hw:
        fs = frame(oracle_hw[0], oracle_hw[1], 5)
        o = eng.render(fs)
        ref = pipeline.render(fs, onnx_exec.Executor(path).run, batch=1, tile=tile, scaling=scale, overlap=(0.0625, 0.0625), tta=tta, net_dtype=np.float16)
        d = np.abs(o.astype(int) - ref.astype(int))
        mse = float(np.mean(d.astype(np.float64) ** 2)); psnr = 99.0 if mse == 0 else 10 * np.log10(255 ** 2 / mse)
        msg += f"; oracle {oracle_hw[1]}x{oracle_hw[0]}: max LSB diff {d.max()}, PSNR {psnr:.1f} dB"
    print(msg, flush=True)
    eng.close()

def graph_delta(tag, model, scale, noise, batch, tile, hw):
    """hipGraph replay of the network passes against plain launches (W2X_NO_GRAPH, read at load): resident ms per frame and the
    wall time of render() calls, same engine file, same frame; outputs must be identical."""
    path = sm.model_path(os.path.join(work, tag), model, scale, noise)
    if not os.path.exists(path):
        sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise), path, 1, tile, dynamic=True)
    f = frame(hw[0], hw[1], 3)
    res = {}
    for mode in ("graph", "plain"):
        if mode == "plain": os.environ["W2X_NO_GRAPH"] = "1"
        else: os.environ.pop("W2X_NO_GRAPH", None)
        eng = pkg.Img2Img()
        assert eng.build(path, pkg.BuildConfig.fixed(batch, tile)), eng.last_error()
        assert eng.load(path, pkg.RenderConfig(batchSize=batch, height=tile, width=tile, scaling=scale)), eng.last_error()
        out = eng.render(f); eng.render(f); eng.render(f)          # eager, capture, replay
        ms = eng.bench_resident(5)
        t0 = time.perf_counter()
        for _ in range(5): eng.render(f)
        wall = (time.perf_counter() - t0) / 5 * 1e3
        n = pkg.calculate_tiles(hw[1], hw[0], hw[1] * scale, hw[0] * scale, tile, eng.output_tile_size, scale, (0.0625, 0.0625))[0]
        res[mode] = (ms, wall, out, n, eng.pass_tiles)
        eng.close()
    os.environ.pop("W2X_NO_GRAPH", None)
    g_, p_ = res["graph"], res["plain"]
    print(f"{tag}: {model} s{scale} B{batch} T{tile} {hw[1]}x{hw[0]}: {g_[3]} tiles in passes of {g_[4]}: hipGraph replay {g_[0]:.2f} ms/frame resident "
          f"({g_[1]:.2f} ms per render() call) vs plain launches {p_[0]:.2f} ms ({p_[1]:.2f} ms per call); identical output: {np.array_equal(g_[2], p_[2])}", flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "4", "5", "g"]
    if "g" in which:
        graph_delta("graph_t64", "swin_unet/art", 4, 3, 1, 64, (1080, 1920))      # S = 1: one tile per pass, ~40 launches per tile
        graph_delta("graph_cfg5", "swin_unet/art_scan", 4, 3, 16, 640, (2160, 3840))
    if "2" in which: run("cfg2", "cunet/art", 2, 1, 4, 256, (1080, 1920), oracle_hw=(200, 300))
    if "4" in which: run("cfg4", "swin_unet/photo", 4, 3, 8, 400, (1080, 1920), tta=True, oracle_hw=(100, 380))
    if "5" in which: run("cfg5", "swin_unet/art_scan", 4, 3, 16, 640, (2160, 3840), oracle_hw=(120, 600))
