"""CPU-side checks of the product library: it loads, exports every symbol of include/w2x/c_api.h, and its host
logic (tile grid, ramps, hash, ONNX lowering, error convention) agrees with the oracle.  No compute calls."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from oracle import onnx_exec, pipeline as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "w2x", "c_api.h")).read()
    body = hdr[hdr.index('extern "C"'):]
    declared = set(re.findall(r"\b(w2x_[a-z0-9_]+)\s*\(", body))
    declared -= {"w2x_message_fn", "w2x_progress_fn"}
    assert len(declared) >= 15
    L = ctypes.CDLL(pkg.lib_path)
    for name in sorted(declared):
        assert hasattr(L, name), f"libw2x.so does not export {name}"
    from importlib import import_module
    eng = import_module("waifu2x-tensorrt_amd.engine")
    assert declared == set(eng.EXPORTED_SYMBOLS)


@pytest.mark.parametrize("W,H", [(1920, 1080), (256, 256), (300, 200), (64, 64), (3840, 2160), (17, 999)])
@pytest.mark.parametrize("T,s,Tout", [(64, 2, 56), (256, 2, 440), (256, 4, 960), (400, 4, 1536), (640, 4, 2496), (128, 1, 72), (64, 1, 48)])
@pytest.mark.parametrize("ov", [0.125, 0.0625, 0.03125, 0.0])
def test_tile_grid_matches_oracle(pkg, W, H, T, s, Tout, ov):
    n, ins, outs = pkg.calculate_tiles(W, H, W * s, H * s, T, Tout, s, (ov, ov))
    n2, ins2, outs2 = P.calculate_tiles(W, H, W * s, H * s, (T, T), (Tout, Tout), s, (ov, ov))
    assert n == n2
    assert np.array_equal(ins, np.array([r.astuple() for r in ins2], np.int32).reshape(-1, 4))
    assert np.array_equal(outs, np.array([r.astuple() for r in outs2], np.int32).reshape(-1, 4))


@pytest.mark.parametrize("ov,size", [(64, 960), (32, 440), (8, 56), (100, 1536), (0, 48)])
def test_tile_weights_bit_exact(pkg, ov, size):
    ref = P.create_tile_weights((ov, ov), (size, size))
    for which in range(4):
        got = pkg.tile_weights(which, ov, ov, size)
        assert np.array_equal(got.view(np.uint32), ref[which].view(np.uint32))


def test_sha256(pkg):
    for msg in (b"", b"abc", b"AMDInstinctMI355X.FP16.4.4.4.3.3.3.256.256.256.256.256.256", os.urandom(1000)):
        assert pkg.sha256_hex(msg) == hashlib.sha256(msg).hexdigest()


@pytest.mark.parametrize("model,scale,tile,small", [("cunet/art", 2, 64, False), ("swin_unet/art", 4, 64, True), ("swin_unet/art", 2, 40, True)])
def test_lowering_covers_graph_and_counts_flops(pkg, onnx_model, model, scale, tile, small):
    path = onnx_model(model, scale, 2, tile, small=small)
    d = pkg.describe_plan(path, 2, tile)
    flops = int(re.search(r"flops=(\d+)", d).group(1))
    assert flops == onnx_exec.count_flops(path, (2, 3, tile, tile))["total"]
    import synth_models as sm
    to = sm.output_tile_size(model, scale, tile)
    assert f"out=[2,3,{to},{to}]" in d


@pytest.mark.parametrize("model,scale,small", [("swin_unet/art", 4, True), ("swin_unet/art", 4, False), ("cunet/art", 2, False)])
def test_opset13_export_with_decomposed_layernorm_lowers_to_the_same_plan(pkg, onnx_model, model, scale, small):
    """nvonnxparser takes whatever opset the file was written with (img2img_build.cpp:81-88).  Below opset 17 LayerNorm arrives as
    ReduceMean / Sub / Pow / ReduceMean / Add / Sqrt / Div / Mul / Add; the lowering must fold that chain exactly like the
    LayerNormalization node: same fused ops, same FLOP count (names aside), and the oracle agrees between the two files."""
    p13, p17 = onnx_model(model, scale, 2, 64, small=small, opset=13), onnx_model(model, scale, 2, 64, small=small, opset=17)
    strip = lambda d: [re.sub(r"\[[^\]]*\]$", "", line) for line in d.splitlines()[2:]]
    d13, d17 = pkg.describe_plan(p13, 2, 64), pkg.describe_plan(p17, 2, 64)
    assert strip(d13) == strip(d17)
    assert re.search(r"flops=(\d+)", d13).group(1) == re.search(r"flops=(\d+)", d17).group(1)
    if small or model.startswith("cunet"):
        x = np.random.default_rng(0).random((2, 3, 64, 64), dtype=np.float32)
        assert np.abs(onnx_exec.Executor(p13).run(x) - onnx_exec.Executor(p17).run(x)).max() < 1e-5


def test_lowering_rejects_wrong_shape_and_garbage(pkg, onnx_model, tmp_path):
    import synth_models as sm
    path = sm.export_onnx(sm.make_model("cunet/art", 2), str(tmp_path / "static.onnx"), 2, 64, dynamic=False)
    with pytest.raises(pkg.W2xError):
        pkg.describe_plan(path, 3, 64)          # static batch 2 in the file
    assert "in=[5,3,64,64]" in pkg.describe_plan(onnx_model("cunet/art", 2, 2, 64), 5, 64)   # dynamic batch axis: any batch
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x00\x01garbage")
    with pytest.raises(pkg.W2xError):
        pkg.describe_plan(str(bad), 1, 64)
    with pytest.raises(pkg.W2xError):
        pkg.describe_plan(str(tmp_path / "missing.onnx"), 1, 64)


def test_error_convention_without_gpu_or_engine(pkg, onnx_model, tmp_path):
    """bool return + "[function@line] message" through the message callback (logger.h:8, logger.cpp:19-22)."""
    eng = pkg.Img2Img()
    seen = []
    eng.setMessageCallback(lambda sev, msg: seen.append((sev, msg)))
    frame = np.zeros((8, 8, 3), np.uint8)
    out = np.zeros((32, 32, 3), np.uint8)
    assert eng.render(frame, out) is False            # render before load
    assert seen and seen[-1][0] == pkg.Severity.error and re.match(r"\[render@\d+\] ", seen[-1][1])
    cfg = pkg.RenderConfig(batchSize=1, height=64, width=64, scaling=2)
    assert eng.load(str(tmp_path / "nope.onnx"), cfg) is False
    assert re.match(r"\[load@\d+\] ", seen[-1][1])
    eng.close()


@pytest.mark.parametrize("W,H,T,s,Tout,ov", [(1920, 1080, 256, 4, 960, 0.0625), (1920, 1080, 256, 2, 440, 0.0625), (3840, 2160, 640, 4, 2496, 0.0625),
                                             (1920, 1080, 256, 4, 960, 0.0), (300, 200, 64, 2, 56, 0.125), (1920, 1080, 400, 4, 1536, 0.03125)])
@pytest.mark.parametrize("parts", [1, 2, 3, 8, 16])
def test_strip_plan_partitions_a_frame(pkg, W, H, T, s, Tout, ov, parts):
    """Multi-GPU split of one frame: the strips' output column ranges tile [0, W*s) without gaps or overlap, each
    strip's tiles are a contiguous range of the reference's tile order, and every tile that covers one of its pixels
    (per the oracle's rects) is inside that range - so composing a strip never needs another device's tiles."""
    n, _, outs = P.calculate_tiles(W, H, W * s, H * s, (T, T), (Tout, Tout), s, (ov, ov))
    xs = []
    for part in range(parts):
        first, cnt, x0, x1 = pkg.strip_plan(W, H, W * s, H * s, T, Tout, s, (ov, ov), part, parts)
        if cnt == 0:
            assert x0 == x1 == 0
            continue
        assert 0 <= first and first + cnt <= n and x0 < x1
        xs.append((x0, x1))
        for t, r in enumerate(outs):
            covers = r.x < x1 and r.x + r.w > x0
            assert (not covers) or first <= t < first + cnt, (part, t, r.astuple())
    xs.sort()
    assert xs[0][0] == 0 and xs[-1][1] == W * s
    assert all(a[1] == b[0] for a, b in zip(xs, xs[1:]))
    if parts == 1:
        assert pkg.strip_plan(W, H, W * s, H * s, T, Tout, s, (ov, ov), 0, 1) == (0, n, 0, W * s)


@pytest.mark.parametrize("model,scale,tile,small", [("cunet/art", 2, 64, False), ("swin_unet/art", 4, 64, True), ("swin_unet/art", 4, 64, False)])
def test_engine_file_is_validated_on_read(pkg, onnx_model, tmp_path, model, scale, tile, small):
    """The plan file is an input from disk (img2img_load.cpp:137-154): a truncated or edited file must be refused with a
    reason, never followed into tensors[] / blobs[] or onto the device.  Host-only halves of build() / load()."""
    path = onnx_model(model, scale, 2, tile, small=small)
    eng = str(tmp_path / "plan.w2x")
    assert pkg.write_engine_file(path, 2, tile, eng)
    ok, why = pkg.validate_engine_file(eng)
    assert ok, why
    good = open(eng, "rb").read()
    bad = str(tmp_path / "bad.w2x")
    # truncations at many points
    for cut in (0, 7, 12, 40, len(good) // 3, len(good) // 2, len(good) - 9, len(good) - 1):
        open(bad, "wb").write(good[:cut])
        ok, why = pkg.validate_engine_file(bad)
        assert not ok and why
    # flipped bytes: header fields, and a sweep over the op records at the end of the file (tensor / blob ids, shapes).  Every
    # outcome must be a clean verdict; edits that hit an id or a shape must be refused.
    rng = np.random.default_rng(7)
    refused = 0
    tail = max(64, len(good) - 20000)
    for pos in list(range(16, 64, 4)) + [int(v) for v in rng.integers(tail, len(good) - 8, 300)]:
        b = bytearray(good)
        b[pos:pos + 4] = (0x7fffff00).to_bytes(4, "little")
        open(bad, "wb").write(bytes(b))
        ok, why = pkg.validate_engine_file(bad)
        refused += (not ok)
    assert refused >= 50
    # sizes the kernels index with.  Every tensor an op touches (describe_plan: refs=...) must keep the plan's B tile slots
    # (per-image strides, tile-group addressing of the arena) and may not shrink in H, W, C or element size behind the ops' backs:
    # the squeeze-excite pool / gate tensors, the attention qkv / out rows, the LayerNorm statistics side tensors, the gated maps.
    import re
    import struct
    desc = pkg.describe_plan(path, 2, tile)
    refs = sorted({int(t) for m in re.finditer(r"refs=([t0-9,]+)", desc) for t in re.findall(r"t(\d+)", m.group(1))})
    assert len(refs) >= 20
    off = 68                                                       # magic, version, 3 struct sizes, 9 header ints, flops (plan.cpp serialize)
    off += 8 + struct.unpack_from("<Q", good, off)[0]              # model kind
    nt = struct.unpack_from("<Q", good, off)[0]; off += 8
    assert max(refs) < nt
    edits = 0
    for k in refs:
        for f, name in enumerate("BHWCe"):
            v = struct.unpack_from("<i", good, off + 20 * k + 4 * f)[0]
            for nv in ([v + 1, v // 2] if f == 0 else [6 - v] if f == 4 else [v // 2]):
                if nv == v or nv <= 0:
                    continue
                b = bytearray(good); struct.pack_into("<i", b, off + 20 * k + 4 * f, nv)
                open(bad, "wb").write(bytes(b))
                ok, why = pkg.validate_engine_file(bad)
                assert not ok and why, (k, name, v, nv)
                edits += 1
    assert edits >= 100


def test_infer_validates_the_blob_shape(pkg):
    """w2x_infer reads batchSize*3*T*T floats, so the binding must refuse any other shape (img2img_infer.cpp:43-68)."""
    eng = pkg.Img2Img()
    with pytest.raises(pkg.W2xError):
        eng.infer(np.zeros((1, 3, 64, 64), np.float32))
    eng._batch, eng._tile = 2, 64           # as load() would record them
    with pytest.raises(ValueError):
        eng.infer(np.zeros((1, 3, 64, 64), np.float32))
    with pytest.raises(ValueError):
        eng.infer(np.zeros((2, 3, 32, 64), np.float32))
    eng.close()


@pytest.mark.parametrize("model,scale", [("cunet/art", 2), ("swin_unet/art", 4)])
def test_tf32_requests_lower_to_an_fp32_plan_of_unfused_ops(pkg, onnx_model, model, scale):
    """Precision::TF32 (config.h:7-10) has no matrix instruction on gfx950; such a build gets the fp32 engine: the same graph
    lowered with fp32 activations, weights and bias tables and none of the fused (fp16) transformer kernels.  The plan must
    survive the engine-file round trip (describe_plan serialises and validates it)."""
    path = onnx_model(model, scale, 2, 64, noise=1)
    d16, d32 = pkg.describe_plan(path, 2, 64), pkg.describe_plan(path, 2, 64, pkg.Precision.TF32)
    assert "precision=fp16" in d16.splitlines()[0] and "precision=fp32" in d32.splitlines()[0]
    assert pkg.describe_plan(path, 2, 64, pkg.Precision.FP32) == d32      # FP32 (exact products) and TF32 (split-bf16 products) run one plan; the choice is made at launch
    assert " swinattn " not in d32 and " mlp " not in d32
    if model.startswith("swin"):
        assert " swinattn " in d16 and " attn heads=" in d32
    act = lambda d: int(re.search(r"activation_bytes=(\d+)", d).group(1))
    flops = lambda d: int(re.search(r"flops=(\d+)", d).group(1))
    assert flops(d16) == flops(d32)                       # the same algorithmic work
    assert act(d32) > 1.5 * act(d16) or model.startswith("swin")   # fp32 maps (the un-fused swin plan also keeps qkv / hidden maps)


def test_cunet_gates_are_folded_into_their_consumers(pkg, onnx_model, monkeypatch):
    """No in-place scaling pass is left in a cunet plan: each squeeze-excite gate rides on the operand load of the (transposed)
    convolution that consumes the map and on the skip add that reads it; the debug switch no_se_fold and fp32 plans keep the pass."""
    path = onnx_model("cunet/art", 2, 2, 64, noise=1)
    d = pkg.describe_plan(path, 2, 64)
    assert " scale t" not in d and d.count(" a*gate") == 4 and d.count(" res*gate") == 1
    assert pkg.describe_plan(path, 2, 64, pkg.Precision.TF32).count(" scale t") == 4
    with pkg.debug_switches(no_se_fold=1):
        d = pkg.describe_plan(path, 2, 64)
    assert d.count(" scale t") == 4 and "gate" not in d
    with pytest.raises(pkg.W2xError):
        pkg.debug_switches(no_such_switch=1).__enter__()


@pytest.mark.parametrize("W,H,T,s,Tout,ov", [(1920, 1080, 256, 4, 960, 0.0625), (1920, 1080, 256, 2, 440, 0.0625), (3840, 2160, 640, 4, 2496, 0.0625),
                                             (1920, 1080, 256, 4, 960, 0.0), (300, 200, 64, 2, 56, 0.125), (1920, 1080, 400, 4, 1536, 0.03125), (100, 1000, 64, 4, 192, 0.0625)])
@pytest.mark.parametrize("parts", [1, 2, 3, 5, 8, 16, 64])
def test_shard_plan_partitions_tiles_and_canvas(pkg, W, H, T, s, Tout, ov, parts):
    """The every-tile-once split of one frame (SURVEY 8e second option, w2x_shard_plan): the parts' tile ranges partition the reference's
    tile order, their rectangles partition the canvas, and every tile that covers a pixel of a part's rectangles (per the oracle's rects)
    is one of its own or one of the tiles [halo_first, first) in front of them - what the seam exchange brings."""
    n, _, outs = P.calculate_tiles(W, H, W * s, H * s, (T, T), (Tout, Tout), s, (ov, ov))
    canvas = np.zeros((H * s // 4 + 1, W * s // 4 + 1), np.int32)            # coverage counted on a 4-pixel lattice (rect edges are multiples of the stride)
    nxt = 0
    sizes = []
    for part in range(parts):
        first, cnt, halo, rects = pkg.shard_plan(W, H, W * s, H * s, T, Tout, s, (ov, ov), part, parts)
        if cnt == 0:
            assert not rects
            continue
        assert first == nxt and 0 <= halo <= first and len(rects) <= 3
        nxt = first + cnt
        sizes.append(cnt)
        for x, y, w, h in rects:
            assert w > 0 and h > 0 and x >= 0 and y >= 0 and x + w <= W * s and y + h <= H * s
            canvas[-(-y // 4):-(-(y + h) // 4), -(-x // 4):-(-(x + w) // 4)] += 1
            for t, r in enumerate(outs):                                     # tiles that reach into the rectangle
                if r.x < x + w and r.x + r.w > x and r.y < y + h and r.y + r.h > y:
                    assert halo <= t < first + cnt, (part, t, (x, y, w, h), r.astuple())
    assert nxt == n
    assert (canvas[:-(-H * s // 4), :-(-W * s // 4)] == 1).all()
    assert max(sizes) - min(sizes) <= 1 or parts > n                          # balanced to within a tile
    if parts == 1:
        assert pkg.shard_plan(W, H, W * s, H * s, T, Tout, s, (ov, ov), 0, 1) == (0, n, 0, [(0, 0, W * s, H * s)])


def test_shard_split_bound_table(pkg, capsys):
    """What ONE image can gain from N GPUs when every tile is computed once (w2x_render_sharded) next to the whole-column strips of
    test_strip_split_redundancy_table: the largest part bounds the speed-up; the seam exchange moves two blend bands per tile in front of a part."""
    import synth_models as sm
    cfgs = {"configs[1] cunet/art s2 T256 1080p": ("cunet/art", 2, 256, 1920, 1080), "configs[2] swin_unet/art s4 T256 1080p": ("swin_unet/art", 4, 256, 1920, 1080),
            "configs[3] swin_unet/photo s4 T400 1080p": ("swin_unet/photo", 4, 400, 1920, 1080), "configs[4] swin_unet/art_scan s4 T640 2160p": ("swin_unet/art_scan", 4, 640, 3840, 2160)}
    want = {"configs[1] cunet/art s2 T256 1080p": {2: 30, 4: 15, 8: 8}, "configs[2] swin_unet/art s4 T256 1080p": {2: 23, 4: 12, 8: 6},
            "configs[3] swin_unet/photo s4 T400 1080p": {2: 9, 4: 5, 8: 3}, "configs[4] swin_unet/art_scan s4 T640 2160p": {2: 14, 4: 7, 8: 4}}
    lines = []
    for name, (m, s, T, W, H) in cfgs.items():
        To = sm.output_tile_size(m, s, T)
        ov = (0.0625, 0.0625)
        n, _, outs = pkg.calculate_tiles(W, H, W * s, H * s, T, To, s, ov)
        ovpx = int(outs[1][1] and (To - (outs[1][1] - outs[0][1])))          # blend band in output pixels (tile 1 sits below tile 0)
        for N in (2, 4, 8):
            parts = [pkg.shard_plan(W, H, W * s, H * s, T, To, s, ov, p, N) for p in range(N)]
            assert sum(c for _, c, _, _ in parts) == n
            largest = max(c for _, c, _, _ in parts)
            assert largest == want[name][N], (name, N, largest)
            halo = max(f - h for f, c, h, _ in parts)
            mb = halo * 2 * ovpx * To * 8 / 1e6
            lines.append(f"{name}: N={N}: {n} tiles, largest part {largest} -> speed-up <= {n / largest:.2f}x; seam exchange <= {halo} tiles x 2 bands = {mb:.1f} MB per part")
    with capsys.disabled():
        print("\n" + "\n".join(lines))


def test_torchvision_operator_order_lowers_onto_the_same_plan(pkg, onnx_model):
    """The release graphs were exported from nunif's swin_unet, whose blocks most likely trace torchvision's shifted_window_attention (neither is available
    offline).  Its operator order differs from the synthetic graphs' in three places the pattern matcher must see through: F.pad to the window multiple in
    front (a Pad node with all-zero pads here), the shift mask built inside the traced function (new_zeros, nine slice assignments, view / permute, a
    difference of two unsqueezes, two masked_fill - all constant at a static shape), and x[:, :H, :W, :] behind the reverse roll (one whole-map Slice per
    axis).  Same weights exported both ways: the CPU oracle gives identical outputs, and the loader lowers both files onto the same fused plan."""
    import re
    from oracle import onnx_exec
    std, tv = onnx_model("swin_unet/art", 4, 1, 64, noise=2), onnx_model("swin_unet/art", 4, 1, 64, noise=2, variant={"tv": 1})
    x = np.random.default_rng(3).random((1, 3, 64, 64), dtype=np.float32)
    assert np.array_equal(onnx_exec.Executor(std).run(x), onnx_exec.Executor(tv).run(x))
    strip = lambda d: [re.sub(r" \[[^\]]*\]$", "", l) for l in d.splitlines()]
    a, b = strip(pkg.describe_plan(std, 1, 64)), strip(pkg.describe_plan(tv, 1, 64))
    assert a == b and sum("swinattn" in l for l in a) == 14 and sum(l.split()[1:2] == ["mlp"] for l in a) == 14


def test_strip_split_redundancy_table(pkg, capsys):
    """Single-frame mode: strip p owns whole tile columns and recomputes the neighbouring column whose
    blend band reaches into its pixels, so the tiles rendered over all strips exceed the frame's tiles and the largest strip bounds
    the speed-up; the strips are balanced by the columns they render (own + recomputed).  The table below (printed, and kept in DESIGN.md section 7) is what an N-GPU run of ONE image can reach at best;
    streams shard by frame instead (no redundancy)."""
    import synth_models as sm
    cfgs = {"configs[1] cunet/art s2 T256 1080p": ("cunet/art", 2, 256, 1920, 1080), "configs[2] swin_unet/art s4 T256 1080p": ("swin_unet/art", 4, 256, 1920, 1080),
            "configs[3] swin_unet/photo s4 T400 1080p": ("swin_unet/photo", 4, 400, 1920, 1080), "configs[4] swin_unet/art_scan s4 T640 2160p": ("swin_unet/art_scan", 4, 640, 3840, 2160)}
    want = {"configs[1] cunet/art s2 T256 1080p": {2: (66, 36), 4: (78, 24), 8: (102, 18)}, "configs[2] swin_unet/art s4 T256 1080p": {2: (50, 25), 4: (60, 15), 8: (80, 10)},
            "configs[3] swin_unet/photo s4 T400 1080p": {2: (21, 12), 4: (27, 9), 8: (33, 6)}, "configs[4] swin_unet/art_scan s4 T640 2160p": {2: (32, 16), 4: (40, 12), 8: (52, 8)}}
    lines = []
    for name, (m, s, T, W, H) in cfgs.items():
        To = sm.output_tile_size(m, s, T)
        ov = (0.0625, 0.0625)
        n = pkg.calculate_tiles(W, H, W * s, H * s, T, To, s, ov)[0]
        for N in (2, 4, 8):
            parts = [pkg.strip_plan(W, H, W * s, H * s, T, To, s, ov, p, N) for p in range(N)]
            total, largest = sum(c for _, c, _, _ in parts), max(c for _, c, _, _ in parts)
            xs = sorted((x0, x1) for _, c, x0, x1 in parts if c)
            assert xs[0][0] == 0 and xs[-1][1] == W * s and all(a[1] == b[0] for a, b in zip(xs, xs[1:]))     # the strips tile the output columns
            assert (total, largest) == want[name][N], (name, N, total, largest)
            lines.append(f"{name}: N={N}: {total} tiles rendered for {n} (+{100 * (total - n) // n} %), largest strip {largest} -> speed-up <= {n / largest:.2f}x")
    with capsys.disabled():
        print("\n" + "\n".join(lines))


def _export(net, path, shape, opset=17):
    import torch
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    import warnings
    net.eval()
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        torch.onnx.export(net, (torch.zeros(*shape),), path, dynamo=False, opset_version=opset, input_names=["x"], output_names=["y"], do_constant_folding=True)


def test_graphs_the_loader_does_not_take_fail_naming_the_node(pkg, tmp_path):
    """img2img_build.cpp:81-88 hands any ONNX file to the parser; what this loader cannot lower must end as a build error that names
    the node (like TensorRT's parser on an unsupported layer), never as a wrong plan: a zero-padded 3x3 convolution, a transposed
    convolution with output_padding, a positive Pad, an operator outside the set."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class Padded(nn.Module):
        def __init__(self): super().__init__(); self.c = nn.Conv2d(3, 8, 3, 1, 1); self.d = nn.Conv2d(8, 3, 3, 1, 0)
        def forward(self, x): return self.d(F.leaky_relu(self.c(x), 0.1))

    class OutPad(nn.Module):
        def __init__(self): super().__init__(); self.c = nn.Conv2d(3, 8, 3, 1, 0); self.t = nn.ConvTranspose2d(8, 3, 3, 2, 0, output_padding=1)
        def forward(self, x): return self.t(F.leaky_relu(self.c(x), 0.1))

    class PosPad(nn.Module):
        def __init__(self): super().__init__(); self.c = nn.Conv2d(3, 3, 3, 1, 0)
        def forward(self, x): return self.c(F.pad(x, (2, 2, 2, 2), mode="reflect"))

    class Odd(nn.Module):
        def __init__(self): super().__init__(); self.c = nn.Conv2d(3, 3, 3, 1, 0)
        def forward(self, x): return torch.atan(self.c(x))

    # transformer shapes outside the kernels' range: rows of 256 channels under a LayerNorm (base width 128), windows of 12 x 12
    import synth_models as sm
    for name, kw, needle in (("c128", {"heads": 8, "base_dim": 128}, "LayerNorm over rows of 256 channels"), ("ws12", {"ws": 12}, "144 tokens per window")):
        path = str(tmp_path / name / "m.onnx")
        sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=3, variant=kw), path, 1, 64, dynamic=True)
        with pytest.raises(pkg.W2xError) as ei:
            pkg.describe_plan(path, 1, 64)
        assert needle in str(ei.value) and "/swin" in str(ei.value), str(ei.value)
    for name, net, needle in (("padded", Padded(), "Conv"), ("outpad", OutPad(), "ConvTranspose"), ("pospad", PosPad(), "Pad"), ("odd", Odd(), "Atan")):
        path = str(tmp_path / f"{name}.onnx")
        _export(net, path, (1, 3, 32, 32))
        with pytest.raises(pkg.W2xError) as ei:
            pkg.describe_plan(path, 1, 32)
        msg = str(ei.value)
        assert needle in msg, (name, msg)
