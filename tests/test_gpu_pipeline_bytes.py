"""The byte / index half of the path against the ORACLE, bit for bit, on the GPU.

Every other frame comparison with the oracle lets the network's fp16 tolerance in (<= 1 LSB).  Here the oracle pipeline
(oracle/pipeline.py: calculateTiles, padRoi, the 8 dihedral augmentations and their inverses, the batch fill with zero-pad slots,
the TTA sum and its 1/8, the L,T,R,B ramp products on the clipped rect, the canvas add in tile order, rint(x*255) with saturation,
RGB<->BGR; img2img_render.cpp:7-352, img2img_load.cpp:29-52, img2img_infer.cpp:5-39) is run on the CPU with the ENGINE's own
network as its `net` (Img2Img.infer through the C ABI = the reference's private infer(), img2img_infer.cpp:41-93), so that both
sides see identical network outputs and everything around the network must agree to the byte:

        pipeline.render(frame, net=eng.infer, ...)  ==  eng.render(frame)

This is the tier's bar for integer / byte work.  Network outputs of a tile do not depend on its batch slot or on the tiles it shares
a pass with (asserted bit-exactly in test_gpu_parity.py), which is what lets infer() stand in for the passes render() runs."""
import os

import numpy as np
import pytest

from oracle import onnx_exec, pipeline
from parity_util import frame_report
from test_gpu_parity import make_engine, smooth_frame

pytestmark = pytest.mark.gpu


def noisy_frame(rows, cols, seed):
    """i.i.d. uniform bytes: saturation / clip paths and every rounding tie the smooth frames do not reach."""
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


def oracle_with_engine_net(eng, frame, *, batch, tile, scale, ov, tta=False, bug=False, fp16=True):
    return pipeline.render(frame, eng.infer, batch=batch, tile=tile, scaling=scale, overlap=(ov, ov), tta=tta, tta_bug_compat=bug,
                           net_dtype=np.float16 if fp16 else None, tile_out=eng.output_tile_size)


def assert_same_bytes(tag, out, ref):
    if not np.array_equal(out, ref):
        d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
        nz = np.argwhere(d.any(-1))
        raise AssertionError(f"{tag}: {len(nz)} pixels differ (max {int(d.max())} LSB), rows [{nz[:, 0].min()}, {nz[:, 0].max()}] "
                             f"cols [{nz[:, 1].min()}, {nz[:, 1].max()}], first {nz[:5].tolist()}")


@pytest.mark.parametrize("ov", [0.125, 0.0625, 0.03125, 0.0])
@pytest.mark.parametrize("tta,bug", [(False, False), (True, False), (True, True)])
def test_tile_pipeline_is_byte_exact_against_the_oracle(pkg, onnx_model, ov, tta, bug):
    """Full-width swin_unet graph (the fused kernels of the benchmark) at T=64, batch 3: every blend setting of the CLI
    (main.cpp:110-121), TTA off / on / bug-compatible (Q1, img2img_render.cpp:313-316), a ragged frame whose tile count is not a
    multiple of the batch (partial last batch, zero-pad slots) and whose last row / column of tiles is clipped."""
    path = onnx_model("swin_unet/art", 4, 3, 64)
    eng = make_engine(pkg, path, 3, 64, 4, overlap=(ov, ov), tta=tta, ttaBugCompat=bug)
    for k, frame in enumerate((smooth_frame(101, 139, 3), noisy_frame(75, 50, 4))):
        out = eng.render(frame)
        ref = oracle_with_engine_net(eng, frame, batch=3, tile=64, scale=4, ov=ov, tta=tta, bug=bug)
        assert_same_bytes(f"ov{ov} tta{int(tta)} bug{int(bug)} frame{k}", out, ref)
    eng.close()


@pytest.mark.parametrize("model,scale,batch,tile,tta,shape", [
    ("cunet/art", 2, 4, 64, False, (100, 77)), ("cunet/art", 1, 2, 96, True, (70, 50)), ("swin_unet/photo", 2, 2, 88, True, (80, 75)),
    ("swin_unet/art_scan", 1, 1, 64, False, (60, 64)), ("swin_unet/art", 4, 2, 64, False, (48, 48))])
def test_other_graph_families_and_scales_byte_exact(pkg, onnx_model, model, scale, batch, tile, tta, shape):
    """x1 / x2 / x4 output geometry (T' = 56..192), the cunet border (18) and the swin border (8), single-tile frames."""
    path = onnx_model(model, scale, batch, tile)
    eng = make_engine(pkg, path, batch, tile, scale, overlap=(0.0625, 0.0625), tta=tta)
    frame = noisy_frame(*shape, 9)
    assert_same_bytes(f"{model} s{scale}", eng.render(frame), oracle_with_engine_net(eng, frame, batch=batch, tile=tile, scale=scale, ov=0.0625, tta=tta))
    eng.close()


@pytest.mark.parametrize("tta,batch", [(False, 3), (True, 4)])
def test_render_call_in_pipelined_parts_returns_the_one_part_bytes(pkg, onnx_model, monkeypatch, tta, batch):
    """render() runs a frame of >= 16 tiles as up to four PARTS (engine.cpp renderPart: contiguous tile ranges cut at whole reference batches, each
    part's canvas cells composed and sent to the host while the next part computes; W2X_RENDER_PARTS, read at load).  Every part count must return
    the bytes of the one-part frame - which is the oracle's - for 8- and 16-bit frames, with TTA (a tile = 8 slots: the cut is a whole batch for
    any tile) and with a batch that does not divide the tile count."""
    path = onnx_model("swin_unet/art", 4, batch, 64)
    frame = noisy_frame(262, 300, 77)
    deep = (frame.astype(np.uint16) * 257) ^ np.random.default_rng(5).integers(0, 256, frame.shape, dtype=np.uint16)
    outs = {}
    for parts in ("1", "2", "3", "4"):
        monkeypatch.setenv("W2X_RENDER_PARTS", parts)
        eng = make_engine(pkg, path, batch, 64, 4, tta=tta)
        outs[parts] = (eng.render(frame), eng.render(deep), eng.render(frame))
        if parts == "1":
            assert_same_bytes("one part against the oracle", outs[parts][0], oracle_with_engine_net(eng, frame, batch=batch, tile=64, scale=4, ov=0.0625, tta=tta))
        eng.close()
        assert outs[parts][1].dtype == np.uint16
        assert_same_bytes(f"{parts} parts, second call", outs[parts][2], outs[parts][0])
        assert_same_bytes(f"{parts} parts", outs[parts][0], outs["1"][0])
        assert_same_bytes(f"{parts} parts, 16-bit", outs[parts][1], outs["1"][1])


@pytest.mark.parametrize("tta,batch", [(False, 3), (True, 4)])
def test_render_pipelines_agree_on_random_frame_sizes(pkg, onnx_model, monkeypatch, tta, batch):
    """The machinery around a render() call - parts with a small last one, the upload split at the first part's columns, parts rolling into each other,
    one graph per tile group - against the plain path (one part, joins, W2X_RENDER_PARTS=1 W2X_NO_ROLLING=1) on seeded random frame sizes from one tile
    row to 130 tiles, 8- and 16-bit, each rendered twice (eager, then replayed) and followed by a resident replay: same bytes every time.  With TTA a
    tile is eight slots, so a part boundary may fall on any tile."""
    path = onnx_model("swin_unet/art", 4, batch, 64)
    rng = np.random.default_rng(2024 + batch)
    shapes = [(int(rng.integers(40, 520)), int(rng.integers(40, 640))) for _ in range(5 if tta else 10)] + [(48, 600), (600, 48), (199, 263)]
    frames = [noisy_frame(r, c, 100 + k) for k, (r, c) in enumerate(shapes)]
    deep = [(f.astype(np.uint16) * 257) ^ 0x5A for f in frames[:4]]
    outs = {}
    for plain in (True, False):
        if plain: monkeypatch.setenv("W2X_RENDER_PARTS", "1"); monkeypatch.setenv("W2X_NO_ROLLING", "1")
        else: monkeypatch.delenv("W2X_RENDER_PARTS"); monkeypatch.delenv("W2X_NO_ROLLING")
        eng = make_engine(pkg, path, batch, 64, 4, tta=tta)
        got = []
        for f in frames + deep:
            a = eng.render(f); b = eng.render(f)
            assert np.array_equal(a, b)
            assert eng.bench_resident(3) > 0
            assert np.array_equal(eng.render(f), a)
            got.append(a)
        outs[plain] = got
        eng.close()
    for k, (a, b) in enumerate(zip(outs[True], outs[False])):
        assert_same_bytes(f"frame {k} {(frames + deep)[k].shape} {(frames + deep)[k].dtype}", b, a)


@pytest.mark.parametrize("tta", [False, True])
def test_strips_are_byte_exact_against_the_oracle(pkg, onnx_model, tta):
    """the frame rendered as 2 and 3 tile-column strips (w2x_render_strip), reassembled, against the oracle's whole frame."""
    path = onnx_model("swin_unet/art", 4, 2, 64)
    eng = make_engine(pkg, path, 2, 64, 4, tta=tta)
    frame = smooth_frame(100, 300, 31)
    ref = oracle_with_engine_net(eng, frame, batch=2, tile=64, scale=4, ov=0.0625, tta=tta)
    for parts in (2, 3):
        out = np.full_like(ref, 77)
        for part in range(parts):
            assert eng.render_strip(frame, out, part, parts), eng.last_error()
        assert_same_bytes(f"{parts} strips tta{int(tta)}", out, ref)
    eng.close()


def test_config3_engine_byte_exact_at_300x420(pkg, onnx_model):
    """BASELINE configs[2] (swin_unet/art s4 n3 B4 T256, blend 1/16): the benchmark's engine on a 2 x 2-tile frame (one full batch of
    four 256 x 256 tiles, T' = 960, 64-pixel ramps) and on a 3 x 2-tile one (a partial second batch)."""
    path = onnx_model("swin_unet/art", 4, 4, 256)
    eng = make_engine(pkg, path, 4, 256, 4)
    for shape, seed in (((300, 420), 13), ((260, 500), 14)):
        frame = smooth_frame(*shape, seed)
        assert_same_bytes(f"config 3 {shape}", eng.render(frame), oracle_with_engine_net(eng, frame, batch=4, tile=256, scale=4, ov=0.0625))
    eng.close()


@pytest.mark.parametrize("model,scale,tta", [("swin_unet/art", 4, True), ("cunet/art", 2, False)])
def test_fp32_engine_tile_pipeline_byte_exact(pkg, onnx_model, model, scale, tta):
    """The same identity on the fp32 engine (Precision::TF32 requests): fp32 tiles cross the engine boundary unrounded, as in the
    reference (IO tensors are f32, img2img_load.cpp:230)."""
    path = onnx_model(model, scale, 2, 64, noise=1)
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(2, 64, precision=pkg.Precision.TF32)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(precision=pkg.Precision.TF32, batchSize=2, height=64, width=64, scaling=scale, overlap=(0.0625, 0.0625), tta=tta)), eng.last_error()
    frame = noisy_frame(90, 110, 2)
    assert_same_bytes(f"fp32 {model}", eng.render(frame), oracle_with_engine_net(eng, frame, batch=2, tile=64, scale=scale, ov=0.0625, tta=tta, fp16=False))
    eng.close()


def test_config1_as_written_fp32_b1_t64_256x256(pkg, onnx_model):
    """BASELINE configs[0] exactly as stated: cunet/art scale2 noise0, batch 1, tile 64, fp32, one 256 x 256 image = 11 x 11 = 121
    tiles in 121 batches of one (KAT row 1 of SURVEY 8c: T' = 56, sIn 28, border 18, overlaps 4 / 8).  The fp32 engine against the
    fp32 oracle end to end (no fp16 anywhere: <= 1 LSB, ties of rint only), the same frame byte for byte against the oracle pipeline
    fed with the engine's network, and the step schedule of img2img_render.cpp:246-250 through the progress callback."""
    path = onnx_model("cunet/art", 2, 1, 64, noise=0)
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(1, 64, precision=pkg.Precision.FP32)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(precision=pkg.Precision.FP32, batchSize=1, height=64, width=64, scaling=2, overlap=(0.0625, 0.0625))), eng.last_error()
    assert eng.output_tile_size == 56
    n, rin, rout = pkg.calculate_tiles(256, 256, 512, 512, 64, 56, 2, (0.0625, 0.0625))
    assert n == 121 and tuple(rin[0]) == (-18, -18, 64, 64) and tuple(rin[-1]) == (222, 222, 64, 64) and tuple(rout[-1]) == (480, 480, 32, 32)
    frame = smooth_frame(256, 256, 1)
    prog = []
    eng.setProgressCallback(lambda c, t, s: prog.append((c, t)))
    out = eng.render(frame)
    assert [c for c, _ in prog] == list(range(1, 122)) and all(t == 121 for _, t in prog)
    eng.setProgressCallback(None)
    assert_same_bytes("configs[0] pipeline", out, oracle_with_engine_net(eng, frame, batch=1, tile=64, scale=2, ov=0.0625, fp16=False))
    ref = pipeline.render(frame, onnx_exec.Executor(path).run, batch=1, tile=64, scaling=2, overlap=(0.0625, 0.0625), tile_out=56)
    r = frame_report("configs[0] cunet/art s2 n0 B1 T64 fp32 256x256 (121 tiles) vs the fp32 oracle", out, ref)
    assert r["max_lsb"] <= 1 and r["frac_pixels_off_by_1"] < 1e-3, r
    eng.close()


@pytest.mark.parametrize("fp32", [False, True])
def test_sixteen_bit_frames_byte_exact_and_close_to_the_oracle(pkg, onnx_model, fp32):
    """Extension (16-bit images are a TODO upstream, README.md:88): CV_16UC3 frames through render() - u16 * float(1/65535) into the network,
    sat(rint(x * 65535)) out, everything in between as for 8-bit frames.  The oracle pipeline takes uint16 frames the same way, so the identity
    pipeline.render(frame16, net=eng.infer) == eng.render(frame16) must hold to the last of the 16 bits, and against the oracle's own network the
    frame may differ by the fp16 network tolerance (3 ULP16 of [0.5, 1) = 96 of 65535) on the fp16 engine, by summation order on the fp32 one."""
    path = onnx_model("swin_unet/art", 4, 2, 64, noise=1)
    eng = pkg.Img2Img()
    prec = pkg.Precision.FP32 if fp32 else pkg.Precision.FP16
    assert eng.build(path, pkg.BuildConfig.fixed(2, 64, precision=prec)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=2, height=64, width=64, scaling=4, overlap=(0.0625, 0.0625), tta=not fp32)), eng.last_error()
    rng = np.random.default_rng(12)
    frame = rng.integers(0, 65536, (70, 90, 3), dtype=np.uint16)
    frame[:8, :8] = 0; frame[-8:, -8:] = 65535
    out = eng.render(frame)
    assert out.dtype == np.uint16 and out.shape == (280, 360, 3)
    ref = pipeline.render(frame, eng.infer, batch=2, tile=64, scaling=4, overlap=(0.0625, 0.0625), tta=not fp32, net_dtype=None if fp32 else np.float16, tile_out=eng.output_tile_size)
    assert_same_bytes(f"16-bit frame, {'fp32' if fp32 else 'fp16'} engine", out, ref)
    ex = onnx_exec.Executor(path) if fp32 else onnx_exec.Executor(path, act_dtype="float16")
    want = pipeline.render(frame, ex.run, batch=2, tile=64, scaling=4, overlap=(0.0625, 0.0625), tta=not fp32, net_dtype=None if fp32 else np.float16, tile_out=eng.output_tile_size)
    d = np.abs(out.astype(np.int64) - want.astype(np.int64))
    from parity_util import _record
    _record({"test": f"16-bit frame vs the oracle, {'fp32' if fp32 else 'fp16'} engine", "kind": "frame16", "max_lsb16": int(d.max()), "mean_lsb16": float(d.mean())})
    assert d.max() <= (2 if fp32 else 96), (d.max(), d.mean())
    # an 8-bit frame afterwards on the same engine (the graph cache keys on the sample width) and a mixed-depth call is refused
    f8 = (frame >> 8).astype(np.uint8)
    assert_same_bytes("8-bit after 16-bit", eng.render(f8), pipeline.render(f8, eng.infer, batch=2, tile=64, scaling=4, overlap=(0.0625, 0.0625), tta=not fp32,
                                                                             net_dtype=None if fp32 else np.float16, tile_out=eng.output_tile_size))
    assert eng.render(frame, np.zeros((280, 360, 3), np.uint8)) is False
    eng.close()


@pytest.mark.parametrize("tta,ov", [(False, 0.0625), (True, 0.0625), (False, 0.125), (False, 0.0), (True, 0.03125)])
def test_one_frame_over_several_engines_every_tile_once(pkg, onnx_model, monkeypatch, tta, ov):
    """SURVEY 8e, second option (w2x_render_sharded): N engines in one process, engine k computes the k-th contiguous range of the tile
    order ONCE, the blend bands of the ny + 1 tiles in front of a range are copied from the engines that computed them, each engine composes
    the canvas cells of its own tiles - against the oracle's whole frame and against one engine's render(), byte for byte, for 2, 3 and 8
    engines (more engines than tile rows, ranges that start and end inside a tile column, a range shorter than its halo).  On the one-GPU box
    the engines share the card (W2X_DEVICE_MAP), so the exchange runs as device-to-device copies; across GPUs it is peer copies (unmeasured)."""
    monkeypatch.setenv("W2X_DEVICE_MAP", ",".join(["0"] * 8))
    path = onnx_model("swin_unet/art", 4, 2, 64)
    frame = smooth_frame(150, 230, 77)                     # 5 x 4 tiles of 64 at blend 1/16
    engs = []
    for dev in range(8):
        e = pkg.Img2Img()
        assert e.build(path, pkg.BuildConfig.fixed(2, 64, device=dev)), e.last_error()
        assert e.load(path, pkg.RenderConfig(deviceId=dev, batchSize=2, height=64, width=64, scaling=4, overlap=(ov, ov), tta=tta)), e.last_error()
        engs.append(e)
    whole = engs[0].render(frame)
    ref = oracle_with_engine_net(engs[0], frame, batch=2, tile=64, scale=4, ov=ov, tta=tta)
    assert_same_bytes(f"render tta{int(tta)} ov{ov}", whole, ref)
    for n in (2, 3, 8):
        out = np.full_like(whole, 55)
        got = pkg.render_sharded(engs[:n], frame, out)
        assert_same_bytes(f"{n} engines tta{int(tta)} ov{ov}", got, ref)
    # a second frame of another size through the same engines (slab and slot buffers are reused), then plain render() again
    frame2 = noisy_frame(70, 300, 5)
    assert_same_bytes("second frame", pkg.render_sharded(engs[:3], frame2), engs[3].render(frame2))
    assert np.array_equal(engs[1].render(frame), whole)
    for e in engs:
        e.close()


def test_one_frame_over_two_processes_through_ipc_handles(pkg, tmp_path):
    """The same split with ONE PROCESS PER GPU (bench.py --mode shards: the launch contract's ranks): rank r computes its tile range into its slab
    (w2x_shard_compute), the ranks exchange hipIpc handles of their slabs over gloo, rank r copies the seam bands out of its predecessors' slabs and writes
    its canvas cells into a frame the ranks share; bench.py itself compares that frame with rank 0's render() and exits non-zero on a difference.  Two ranks
    sharing the box's GPU (W2X_DEVICE_MAP) on a test-sized frame (bench.py --config 0: 63 tiles of 64); the headline frame over two, three and four ranks
    was rehearsed by hand (profiles/r4_final/bench_shards_*ranks_one_gpu.json)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n in (2,):
        env = dict(os.environ, W2X_DEVICE_MAP=",".join(["0"] * n), W2X_BENCH_WORK=str(tmp_path / f"w{n}"))
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "shards", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--config", "0"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["config"]["mode"] == "shards" and line["config"]["bytes_equal_render"] is True
        # the line certifies where it ran: two ranks, ONE distinct GPU (the ranks share the box's card) - a rehearsal, and it says so
        p = line["placement"]
        assert line["n_ranks"] == n and line["n_gpus"] == 1 and p["distinct_gpus"] == 1 and not p["one_gpu_per_rank"] and line["metric"].startswith(f"REHEARSAL ({n} ranks on 1 GPU)")
        assert len({rec["pid"] for rec in p["ranks"]}) == n and len({rec["pci_bus_id"] for rec in p["ranks"]}) == 1 and p["device_map"] == env["W2X_DEVICE_MAP"]
        assert line["config"]["peer_device_ordinals_seen_by_rank0"] == [0] * n
        assert sum(line["config"]["tiles_per_rank"]) == 63 and min(line["config"]["tiles_per_rank"]) > 0


@pytest.mark.gpu
def test_frame_sharded_bench_line_certifies_where_it_ran(tmp_path):
    """bench.py's headline mode (frames round-robin over ranks) with two ranks on the box's ONE GPU (W2X_DEVICE_MAP=0,0) on the test-sized frame: the line carries
    `placement` (both ranks' pids, the one PCI bus id), n_ranks = 2 but n_gpus = 1 and a metric that starts with REHEARSAL - it cannot be read as a two-GPU
    result - and the host-to-host figure with its per-rank spread next to the resident `value`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, W2X_DEVICE_MAP="0,0", W2X_BENCH_WORK=str(tmp_path / "w"))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    p = d["placement"]
    assert d["n_ranks"] == 2 and d["n_gpus"] == 1 and d["metric"].startswith("REHEARSAL (2 ranks on 1 GPU)") and d["scaling"] == "weak"
    assert p["distinct_gpus"] == 1 and not p["one_gpu_per_rank"] and p["device_map"] == "0,0" and len({x["pid"] for x in p["ranks"]}) == 2
    h = d["host_to_host"]
    assert h and h["ms_per_frame"] > 0 and h["per_rank_ms_per_frame"]["min"] <= h["per_rank_ms_per_frame"]["max"] <= h["ms_per_frame"] + 1e-3
    assert d["value"] > 0 and d["config"]["per_rank_resident_ms_per_step"]["max"] <= d["ms_per_step"] + 1e-3
    assert d["roofline"]["bound"] in ("hbm", "mfma") and "cpu_baseline" not in d


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frames", "shards"])
def test_five_engines_in_five_processes_on_one_card(tmp_path, mode):
    """The launch contract's shape at the size this pool allows: one process per rank, each with its own engine, rendezvous over gloo, page-locked frame rings
    and (shards) hipIpc slab handles of every rank opened by its successors - FIVE ranks on the box's one GPU (W2X_DEVICE_MAP=0,0,0,0,0).  Why five and not the
    node's eight: the pool's process guard ends a run that has more than six processes on the card, and the test runner is one of them (it holds the engine
    library of the tests before this one).  What it shows that the two-rank tests do not: five contexts, five sets of four streams, five page-locked rings and
    (shards) up to four opened peer slabs per rank coexist; every rank's frames are the bytes of render() (bench.py raises otherwise), the shared frame of the
    shards mode equals rank 0's render(), and the line says REHEARSAL (5 ranks on 1 GPU).  Reference: one device per process (src/main.cpp:70-74)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 5
    env = dict(os.environ, W2X_DEVICE_MAP=",".join(["0"] * n), W2X_BENCH_WORK=str(tmp_path / "w"))
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--repeats", "1", "--config", "0", "--no-cpu-baseline"]
    if mode == "shards":
        args += ["--mode", "shards"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    p = d["placement"]
    assert d["n_ranks"] == n and d["n_gpus"] == 1 and d["metric"].startswith(f"REHEARSAL ({n} ranks on 1 GPU)")
    assert p["distinct_gpus"] == 1 and not p["one_gpu_per_rank"] and len({x["pid"] for x in p["ranks"]}) == n and len({x["pci_bus_id"] for x in p["ranks"]}) == 1
    assert d["value"] > 0
    if mode == "shards":
        assert d["config"]["bytes_equal_render"] is True and sum(d["config"]["tiles_per_rank"]) == 63 and min(d["config"]["tiles_per_rank"]) > 0
        assert d["config"]["peer_device_ordinals_seen_by_rank0"] == [0] * n
    else:
        h = d["host_to_host"]
        assert h and h["ms_per_frame"] > 0 and h["per_rank_ms_per_frame"]["min"] <= h["per_rank_ms_per_frame"]["max"]
        assert d["config"]["frames_per_rank"] == 2 and d["scaling"] == "weak"
