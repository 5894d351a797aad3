"""The synchronous render() call (host frame in, host frame out: img2img_render.cpp:226-344) at config 3 against the number of parts the frame is
pipelined in (W2X_RENDER_PARTS, engine.cpp renderPart), pageable and page-locked host buffers; one child process per setting, alternating.
    python tools/ab/render_parts.py [rounds]            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
import numpy as np
ROOT = sys.argv[1]
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
def timed(src, dst, n=12):
    for _ in range(4): assert eng.render(src, dst)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.render(src, dst); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], eng.last_render_ms
pg = timed(frame, out)
hsrc = eng.alloc_host((1080, 1920, 3)); hdst = eng.alloc_host((4320, 7680, 3)); hsrc[...] = frame
pl = timed(hsrc, hdst)
assert np.array_equal(out, hdst)
res = eng.bench_resident(10)
print(f"parts={os.environ.get('W2X_RENDER_PARTS', 'default')} cuts={os.environ.get('W2X_RENDER_CUTS', '-')}  render() pageable median {pg[0]:.2f} min {pg[1]:.2f} ms (compute stream {pg[2]:.2f}) | page-locked median {pl[0]:.2f} min {pl[1]:.2f} ms | resident frame {res:.3f} ms | crc {int(out[::97, ::89].astype(np.uint64).sum())}")
'''
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
base = [] if "nobase" in sys.argv[2:] else [(p, "", "") for p in ("1", "2", "3", "4")]
settings = base + [tuple((a + "::").split(":")[:3]) for a in sys.argv[2:] if a != "nobase"]     # extra settings: PARTS[:CUTS[:NAME=VALUE]], e.g. 3:15,30  4::W2X_COPY_STREAM_PRIO=high
for r in range(rounds):
    for parts, cuts, extra in settings:
        env = dict(os.environ, W2X_RENDER_PARTS=parts)
        if cuts: env["W2X_RENDER_CUTS"] = cuts
        if extra: env[extra.split("=")[0]] = extra.split("=")[1]; print(extra, end="  ")
        p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
        print((p.stdout.strip().splitlines() or ["(no output) " + p.stderr[-400:]])[-1], flush=True)
