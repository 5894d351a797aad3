"""GPU parity: the HIP engine, called through the C ABI, against the oracle on the same seeded inputs.

Tolerances, stated here and asserted below (units: tests/parity_util.py; every comparison is printed and recorded):
  * network output [B,3,T',T'] against the oracle in fp16-boundary mode (an fp16 engine modelled layer by layer):
    max |d| <= NET_MAX_ULP16 fp16 ULPs of the output range (2^-11), mean |d| <= NET_MEAN_ABS; against the fp32 oracle the
    same run is asserted too (NET_MAX_ULP16_VS_FP32), so the bound does not rest on a model of the engine alone.  Measured: see
    profiles/r*_*/parity.jsonl; each bound sits 0.5-1 ULP above the largest figure measured for its path.
  * frames (u8) against the oracle pipeline with the fp16-boundary network: <= FRAME_MAX_LSB, PSNR > 50 dB;
  * everything that is integer / byte work (tile order, padding, TTA index maps, blend masks, u8 rounding) is bit-exact
    AGAINST THE ORACLE in tests/test_gpu_pipeline_bytes.py (the oracle pipeline around the engine's own network == render());
    the batch-size, super-batch, strip, sequence, graph-replay and poison tests here compare engine renders with == on bytes."""
import os
import threading

import numpy as np
import pytest

import synth_models as sm
from oracle import cnet, onnx_exec, pipeline
from parity_util import (BLOCK_MEAN_TOL, ULP16, assert_as_accurate_as_ideal_fp16, check_config_fixture, frame_report, network_report, psnr,  # noqa: F401
                         smooth_frame)

pytestmark = pytest.mark.gpu

# fp16 ULPs of [0.5, 1) = 2^-11 each.  north_star asks 1 against TensorRT itself (not checkable offline); what is checkable:
#  * engine vs the FP32 oracle - the engine's own error.  Measured (profiles/r4_final/parity.jsonl, every graph family, T = 64 ... 400): max 1.04-2.52, p99.9 <= 1.57, rms 0.31-0.46;
#    the oracle's per-operator-fp16 model of an ideal fp16 engine measures max 1.04-2.44, rms 0.31-0.44 against the same fp32 values: the engine is as accurate as that model
#    (asserted per case by parity_util.assert_as_accurate_as_ideal_fp16: max within +0.5, rms within 10 %, p99.9 <= 2).
#  * engine vs the fp16-boundary oracle - the DISTANCE BETWEEN TWO fp16 evaluations, each within ~2 ULP16 of the fp32 value: largest where they straddle it
#    (parity.jsonl `straddle_at_worst`: e.g. [-1.6, +1.4] at the one element that measures 3.00 on swin_unet/art T112) and quantised in whole / half ULPs.  Its bound is
#    therefore the sum of the two, not a statement about the engine: 3.5 (measured 1.0-3.0).  Round 3 asserted 3.0 on a measured 3.00.
NET_MAX_ULP16 = 3.5
NET_MAX_ULP16_UNFUSED = 3.5   # the un-fused operator path (48-channel graphs, debug switch no_fuse_attn): measured 2.0-3.0
NET_MAX_ULP16_VS_FP32 = 3.0   # measured <= 2.52 (T = 256: the maximum over 2.7 M outputs)
NET_MEAN_ABS = 2.2e-4    # measured <= 1.7e-4
FRAME_MAX_LSB = 1        # u8; measured 1 on every case
# north_star's "within 1 ULP fp16 per pixel", as tested quantities (round 5): the FRACTION of the network's outputs within 1 and 2 ULP16 of the fp16-boundary oracle and
# of the fp32 oracle at the shipping tile sizes (T = 256 live, T = 400 against the committed oracle windows).  Floors sit just under the measurements
# (profiles/r5_final/parity.jsonl; README.md has the table): a loss of accuracy that stays under the max-bounds above still fails here.
NET_FRAC_FLOORS = {   # tile: (within 1 ULP16 of the fp16 oracle, within 2, within 1 ULP16 of fp32, within 2)
    256: (0.992, 0.9999, 0.975, 0.9999),     # measured 0.9941 / 0.999998 / 0.9779 / 0.999993 (the fp16-boundary oracle itself: 0.9761 of its outputs within 1 ULP16 of fp32)
    400: (0.992, 0.9999, 0.975, 0.9999),     # measured 0.9940 / 0.999999 / 0.9776 / 0.999994
}


def assert_ulp_fractions(r, tile):
    f1, f2, g1, g2 = NET_FRAC_FLOORS[tile]
    assert r["frac_within_1_ulp16"] >= f1 and r["frac_within_2_ulp16"] >= f2, r
    assert r["frac_within_1_ulp16_vs_fp32_oracle"] >= g1 and r["frac_within_2_ulp16_vs_fp32_oracle"] >= g2, r


def oracle16(path):
    return onnx_exec.Executor(path, act_dtype="float16").run


def make_engine(pkg, path, batch, tile, scale, **kw):
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(batch, tile)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=batch, height=tile, width=tile, scaling=scale, **kw)), eng.last_error()
    return eng


@pytest.mark.parametrize("model,scale,batch,tile,small", [
    ("cunet/art", 2, 1, 64, False), ("cunet/art", 2, 3, 96, False), ("cunet/art", 1, 2, 64, False),
    ("swin_unet/art", 4, 2, 64, True), ("swin_unet/art", 4, 1, 64, False), ("swin_unet/photo", 2, 2, 88, False),
    ("swin_unet/art_scan", 1, 1, 64, False), ("swin_unet/art", 4, 1, 112, False),
    # the tile size of configs[1] / configs[2] (BASELINE.json): persistent MLP waves walking several tiles, 120^2 / 60^2 attention levels,
    # the image head folded into the last MLP launch; 2 s of oracle per tile and mode
    ("swin_unet/art", 4, 1, 256, False), ("cunet/art", 2, 1, 256, False)])
def test_network_matches_oracle(pkg, onnx_model, model, scale, batch, tile, small):
    """trt::Img2Img::infer (img2img_infer.cpp:41-93): [B,3,T,T] -> [B,3,T',T'] on the same ONNX weights."""
    path = onnx_model(model, scale, batch, tile, small=small)
    eng = make_engine(pkg, path, batch, tile, scale)
    assert eng.output_tile_size == sm.output_tile_size(model, scale, tile)
    rng = np.random.default_rng(5)
    x = rng.random((batch, 3, tile, tile), dtype=np.float32).astype(np.float16).astype(np.float32)
    x[0, :, :8, :8] = 0.0; x[-1, :, -8:, -8:] = 1.0
    y = eng.infer(x)
    ref32 = onnx_exec.Executor(path).run(x)
    ref16 = oracle16(path)(x)
    assert not np.isnan(y).any()
    # the fp32 reference both bounds below are measured against is the reading of TWO independent executors (torch operators / C++ loops: tests/test_oracle_cnet.py)
    assert float(np.abs(cnet.Executor(path).run(x) - ref32).max()) < 2e-5
    r = network_report(f"network[{model} s{scale} B{batch} T{tile} {'small' if small else 'full'}]", y, ref16, ref32)
    assert r["max_ulp16"] <= (NET_MAX_ULP16_UNFUSED if small else NET_MAX_ULP16) and r["mean_abs"] <= NET_MEAN_ABS, r
    # and against fp32 arithmetic itself: a loss of accuracy that the fp16-boundary oracle happens to share would still fail here
    assert r["max_ulp16_vs_fp32_oracle"] <= NET_MAX_ULP16_VS_FP32 and r["mean_abs_vs_fp32_oracle"] <= NET_MEAN_ABS, r
    # ... and relative to an ideal fp16 engine: no further from fp32 than the fp16-boundary oracle is
    assert_as_accurate_as_ideal_fp16(r)
    if tile in NET_FRAC_FLOORS and model.startswith("swin"):
        assert_ulp_fractions(r, tile)
    # batch items are independent: same tile in slot 0 and slot B-1 gives the same bytes
    if batch > 1:
        x2 = np.repeat(x[:1], batch, axis=0)
        y2 = eng.infer(x2)
        assert np.array_equal(y2[0], y2[-1])
    eng.close()


def test_network_at_shipping_tile_sizes_against_fixture(pkg, onnx_model):
    """configs[3]'s tile (T = 400, T' = 1536) against the COMMITTED oracle output (tests/golden/net_swin_s4_t400.npz from
    make_net_fixture.py: windows of the fp32 and of the fp16-boundary oracle's output for one seeded tile).  Same bounds as above."""
    from golden.make_net_fixture import tile_input
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net_swin_s4_t400.npz"))
    model, scale, noise, tile, seed = str(z["model"]), int(z["scale"]), int(z["noise"]), int(z["tile"]), int(z["seed"])
    path = onnx_model(model, scale, 1, tile, noise=noise)
    eng = make_engine(pkg, path, 1, tile, scale)
    y = eng.infer(tile_input(tile, seed))[0]
    eng.close()
    assert not np.isnan(y).any()
    got = np.stack([y[:, y0:y0 + h, x0:x0 + w] for y0, x0, h, w in z["windows"]])
    r = network_report(f"network[{model} s{scale} B1 T{tile} full, committed oracle windows]", got, z["ref16"].astype(np.float32), z["ref32"])
    assert r["max_ulp16"] <= NET_MAX_ULP16 and r["mean_abs"] <= NET_MEAN_ABS and r["max_ulp16_vs_fp32_oracle"] <= NET_MAX_ULP16_VS_FP32, r
    assert_as_accurate_as_ideal_fp16(r)
    assert_ulp_fractions(r, tile)
    # whole-output checksum: the per-channel sums of all 3 x 1536^2 values against the fp32 oracle's (mean error per element, in ULP16)
    drift = np.abs(y.astype(np.float64).sum(axis=(1, 2)) - z["sum32"]) / y[0].size / ULP16
    print("PARITY " + str({"test": "network T400 whole-output mean drift per channel (ULP16)", "drift": drift.tolist()}), flush=True)
    assert drift.max() <= 0.1, drift


@pytest.mark.parametrize("model,scale,batch,tile,small,ov,tta,shape", [
    ("swin_unet/art", 4, 2, 64, True, 0.0625, False, (90, 130)),
    ("swin_unet/art", 4, 4, 64, True, 0.0, False, (48, 48)),
    ("swin_unet/art", 2, 3, 64, True, 0.125, True, (70, 50)),
    ("swin_unet/art", 4, 8, 64, True, 0.03125, True, (64, 64)),
    ("cunet/art", 2, 4, 64, False, 0.0625, False, (100, 77)),
    ("cunet/art", 1, 2, 96, False, 0.125, False, (70, 50)),
    ("cunet/art", 2, 1, 64, False, 0.0625, True, (30, 34)),
    # the kernels that ship in the benchmark (96 / 192 channels: fused attention + MLP), with TTA, ragged frames, partial batches
    ("swin_unet/art", 4, 4, 64, False, 0.0625, False, (100, 140)),
    ("swin_unet/photo", 4, 8, 64, False, 0.0625, True, (60, 100)),
    ("swin_unet/art_scan", 2, 3, 88, False, 0.125, True, (80, 75)),
])
def test_render_matches_oracle(pkg, onnx_model, model, scale, batch, tile, small, ov, tta, shape):
    """trt::Img2Img::render (img2img_render.cpp:224-352) end to end, ragged frames, partial last batch, TTA, blend."""
    path = onnx_model(model, scale, batch, tile, small=small)
    eng = make_engine(pkg, path, batch, tile, scale, overlap=(ov, ov), tta=tta)
    frame = smooth_frame(shape[0], shape[1], 3)
    prog = []
    eng.setProgressCallback(lambda c, t, s: prog.append((c, t)))
    out = eng.render(frame)
    ref = pipeline.render(frame, oracle16(path), batch=batch, tile=tile, scaling=scale, overlap=(ov, ov), tta=tta,
                          net_dtype=np.float16)
    r = frame_report(f"render[{model} s{scale} B{batch} T{tile} {'small' if small else 'full'} ov{ov} tta{int(tta)} {shape}]", out, ref)
    assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB, r
    # the step schedule of img2img_render.cpp:246-250: tiles x (8 if tta) steps, rounded up to whole batches
    n_tiles = pkg.calculate_tiles(shape[1], shape[0], shape[1] * scale, shape[0] * scale, tile, eng.output_tile_size, scale, (ov, ov))[0]
    assert prog[-1][1] == -(-(n_tiles * (8 if tta else 1)) // batch)
    assert prog and prog[-1][0] == prog[-1][1] and [c for c, _ in prog] == list(range(1, prog[-1][1] + 1))
    # deterministic: a second render gives identical bytes; strided src/dst views work
    assert np.array_equal(out, eng.render(frame))
    big = np.zeros((shape[0], shape[1] + 5, 3), np.uint8); big[:, :shape[1]] = frame
    dst = np.zeros((shape[0] * scale, shape[1] * scale + 7, 3), np.uint8)
    assert eng.render(big[:, :shape[1]], dst[:, :shape[1] * scale]) is True
    assert np.array_equal(dst[:, :shape[1] * scale], out)
    eng.close()


def test_opset13_graph_with_decomposed_layernorm_runs_on_the_fused_kernels(pkg, onnx_model):
    """The same weights exported at opset 13 (LayerNorm as a ReduceMean / Sub / Pow / ... chain) build, lower onto the same fused
    kernels and give the same bytes as the opset-17 file."""
    p13, p17 = onnx_model("swin_unet/art", 4, 2, 64, opset=13), onnx_model("swin_unet/art", 4, 2, 64, opset=17)
    frame = smooth_frame(100, 140, 9)
    outs = []
    for path in (p13, p17):
        eng = make_engine(pkg, path, 2, 64, 4)
        outs.append(eng.render(frame))
        eng.close()
    assert np.array_equal(outs[0], outs[1])
    ref = pipeline.render(frame, oracle16(p13), batch=2, tile=64, scaling=4, overlap=(0.0625, 0.0625), net_dtype=np.float16)
    r = frame_report("render[swin_unet/art s4 B2 T64 full opset13]", outs[0], ref)
    assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB, r


def test_unfused_attention_core_on_full_width_graphs(pkg, onnx_model, monkeypatch):
    """Graphs whose transformer shapes the fused kernels do not cover keep the QKV / proj linears on the general MFMA GEMM and run
    the attention core on k_attn.hip: attn_mfma_kernel (head sizes 8 / 16 / 32 on v_mfma_f32_16x16x16_f16; the 48-channel test
    graphs use head size 8 everywhere in this file) or, under the debug switch attn_valu, the lane-per-query kernel it replaced.  Here the
    full-width graph is forced down that path (debug switch no_fuse_attn at build time, csrc/switches.h): head sizes 16 and 32, shifted windows, all
    mask classes, against the same oracle and bounds as the fused kernels."""
    path = onnx_model("swin_unet/art", 4, 1, 64, noise=1)
    with pkg.debug_switches(no_fuse_attn=1):
        eng = make_engine(pkg, path, 1, 64, 4)
    rng = np.random.default_rng(11)
    x = rng.random((1, 3, 64, 64), dtype=np.float32).astype(np.float16).astype(np.float32)
    y = eng.infer(x)
    r = network_report("network[swin_unet/art s4 B1 T64 full, un-fused attention core]", y, oracle16(path)(x), onnx_exec.Executor(path).run(x))
    assert r["max_ulp16"] <= NET_MAX_ULP16_UNFUSED and r["mean_abs"] <= NET_MEAN_ABS and r["max_ulp16_vs_fp32_oracle"] <= NET_MAX_ULP16_VS_FP32, r
    eng.close()


def test_tta_bug_compat_mode(pkg, onnx_model):
    """Quirk Q1 (img2img_render.cpp:313-316): optional bug-compatible TTA blends the last de-augmented output."""
    path = onnx_model("swin_unet/art", 2, 4, 64, small=True)
    eng = make_engine(pkg, path, 4, 64, 2, overlap=(0.0625, 0.0625), tta=True, ttaBugCompat=True)
    frame = smooth_frame(50, 60, 9)
    out = eng.render(frame)
    ref = pipeline.render(frame, oracle16(path), batch=4, tile=64, scaling=2, overlap=(0.0625, 0.0625), tta=True,
                          tta_bug_compat=True, net_dtype=np.float16)
    r = frame_report("tta_bug_compat[swin small s2 B4 T64]", out, ref)
    assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB, r
    # and it really is a different picture from the intended mean (the default), which the other tests cover
    eng2 = make_engine(pkg, path, 4, 64, 2, overlap=(0.0625, 0.0625), tta=True)
    assert not np.array_equal(out, eng2.render(frame))
    eng.close(); eng2.close()


@pytest.mark.parametrize("small", [True, False])
def test_batch_size_invariance_bit_exact(pkg, onnx_model, small):
    """Tiles are independent units: the frame must not depend on how tiles are grouped into batches."""
    frame = smooth_frame(120, 150, 4)
    outs = []
    for b in (1, 3, 4):
        eng = make_engine(pkg, onnx_model("swin_unet/art", 4, b, 64, small=small), b, 64, 4)
        outs.append(eng.render(frame)); eng.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_error_paths(pkg, onnx_model, tmp_path):
    path = onnx_model("cunet/art", 2, 2, 64)
    eng = pkg.Img2Img()
    # an engine serves only the precision it was built for (isCompatible, img2img_load.cpp:54-66): a TF32 engine (the fp32 engine
    # on gfx950) on disk does not satisfy an FP16 render configuration
    assert eng.build(path, pkg.BuildConfig.fixed(2, 64, precision=pkg.Precision.TF32)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=2)) is False
    assert "could not satisfy render configuration" in eng.last_error()
    assert eng.load(path, pkg.RenderConfig(precision=pkg.Precision.FP32, batchSize=2, height=64, width=64, scaling=2)) is False     # nor an FP32 one: TF32 and FP32 share a plan, not an engine file
    assert "could not satisfy render configuration" in eng.last_error()
    assert eng.build(path, pkg.BuildConfig.fixed(2, 64)), eng.last_error()
    # no engine for this configuration (img2img_load.cpp:111-112)
    assert eng.load(path, pkg.RenderConfig(batchSize=4, height=64, width=64, scaling=2)) is False
    assert "could not satisfy render configuration" in eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=2)), eng.last_error()
    # wrong dst size
    assert eng.render(np.zeros((10, 10, 3), np.uint8), np.zeros((21, 20, 3), np.uint8)) is False
    # engine files were written next to the model with the reference's naming (img2img_build.cpp:151-155)
    d = os.path.dirname(path)
    names = sorted(os.listdir(d))
    stem = os.path.splitext(os.path.basename(path))[0]
    assert any(n.startswith(stem + "_") and n.endswith(".json") for n in names) and any(n.endswith(".w2x") for n in names)
    eng.close()


def test_compatible_but_not_optimized_engine_is_respecialised(pkg, onnx_model):
    """img2img_load.cpp:100-107: load() takes the first optimized engine, else the first compatible one.  An engine built with a
    range (min 1 / opt 2 / max 4 tiles of 64..96) serves a render configuration inside the range that is not its opt shape: the plan is
    specialised again from the same ONNX file (a warning says so) and the frame is the one a dedicated build gives."""
    path = onnx_model("swin_unet/art", 4, 1, 64, noise=2)
    frame = smooth_frame(90, 120, 31)
    eng = pkg.Img2Img()
    bc = pkg.BuildConfig(0, pkg.Precision.FP16, 1, 2, 4, 3, 3, 3, 64, 64, 96, 64, 64, 96)
    assert eng.build(path, bc), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=4, height=64, width=64, scaling=4)), eng.last_error()      # inside the range, not opt
    assert any("compatible with but not optimized" in m for _, m in eng.messages)
    assert eng.pass_tiles % 4 == 0                                          # whole reference batches per network pass
    got = eng.render(frame)
    assert eng.load(path, pkg.RenderConfig(batchSize=8, height=64, width=64, scaling=4)) is False               # outside the range
    assert "could not satisfy render configuration" in eng.last_error()
    eng.close()
    ref = make_engine(pkg, path, 4, 64, 4)          # adds a dedicated (optimized) engine file next to the ranged one
    assert np.array_equal(ref.render(frame), got)
    ref.close()
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=4, height=64, width=64, scaling=4)), eng.last_error()      # now the optimized one wins
    assert not any("not optimized" in m for _, m in eng.messages)
    eng.close()
    bad = pkg.BuildConfig(0, pkg.Precision.FP16, 4, 2, 1, 3, 3, 3, 64, 64, 64, 64, 64, 64)
    assert pkg.Img2Img().build(path, bad) is False                                                               # min > opt


def test_headline_config_properties(pkg, onnx_model):
    """BASELINE config 3 at full size (swin_unet/art s4 n3 B4 T256, 1920x1080, blend 1/16): the oracle needs minutes
    per frame there, so check size-independent properties instead, and oracle parity on a frame with few tiles."""
    path = onnx_model("swin_unet/art", 4, 4, 256)
    eng = make_engine(pkg, path, 4, 256, 4)
    assert eng.output_tile_size == 960
    frame = smooth_frame(1080, 1920, 7)
    out = eng.render(frame)
    assert out.shape == (4320, 7680, 3)
    again = eng.render(frame)
    if not np.array_equal(out, again):                                   # idempotent / deterministic
        nz = np.argwhere((out != again).any(-1))
        raise AssertionError(f"render is not deterministic: {len(nz)} pixels differ, y[{nz[:, 0].min()},{nz[:, 0].max()}] "
                             f"x[{nz[:, 1].min()},{nz[:, 1].max()}] max {np.abs(out.astype(int) - again.astype(int)).max()}")
    # translation consistency: the same 240x240 content at two different tile positions gives the same pixels
    f2 = np.zeros_like(frame); f2[:] = 128
    patch = smooth_frame(224, 224, 11)
    f2[8:232, 8:232] = patch; f2[8 + 448:232 + 448, 8 + 896:232 + 896] = patch    # tile (0,0) and tile (4,2): origins differ by multiples of 224
    o2 = eng.render(f2)
    a = o2[4 * 40:4 * 200, 4 * 40:4 * 200]; b = o2[4 * (40 + 448):4 * (200 + 448), 4 * (40 + 896):4 * (200 + 896)]
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    # small frame (2x2 tiles at T=256) against the oracle
    small = smooth_frame(300, 420, 13)
    o3 = eng.render(small)
    if os.environ.get("W2X_LIVE_ORACLE"):
        ref = pipeline.render(small, oracle16(path), batch=4, tile=256, scaling=4, overlap=(0.0625, 0.0625), net_dtype=np.float16)
        r = frame_report("config3[swin_unet/art s4 B4 T256 300x420]", o3, ref)
        assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB, r
    else:           # the oracle's frame as a committed fixture (tests/golden/make_config_fixtures.py)
        r = check_config_fixture("3", o3)
        assert r["psnr_db"] > 50.0 and r["max_lsb"] <= FRAME_MAX_LSB and r["max_block_mean_diff"] <= BLOCK_MEAN_TOL, r
    eng.close()


@pytest.mark.parametrize("model,scale,small", [("swin_unet/art", 4, True), ("swin_unet/art", 4, False), ("cunet/art", 2, False)])
def test_pad_slots_are_skipped_and_stale_memory_is_never_read(pkg, onnx_model, monkeypatch, model, scale, small):
    """The zero-pad slots of the last batch (img2img_render.cpp:281) are not computed; W2X_POISON (read by load()) turns every
    stale activation into an fp16 NaN before each frame, so any read of a skipped slot would show up in the picture."""
    path = onnx_model(model, scale, 4, 64, small=small)
    frame = smooth_frame(100, 150, 5)                                    # 3x4 = 12 tiles... not a multiple of the pass size
    monkeypatch.delenv("W2X_POISON", raising=False)
    eng = make_engine(pkg, path, 4, 64, scale)
    clean = eng.render(frame)
    eng.close()
    monkeypatch.setenv("W2X_POISON", "1")
    eng = make_engine(pkg, path, 4, 64, scale)
    assert np.array_equal(clean, eng.render(frame))
    assert np.array_equal(clean, eng.render(frame))
    eng.close()


@pytest.mark.parametrize("model,scale,tile,tta,shape,small", [("swin_unet/art", 4, 64, False, (150, 330), True), ("cunet/art", 2, 64, True, (100, 260), False),
                                                              ("swin_unet/art", 4, 64, True, (100, 300), False)])
def test_strips_reassemble_the_frame_bit_exactly(pkg, onnx_model, model, scale, tile, tta, shape, small):
    """one frame split over N devices by tile-column strips.  Rendering the strips one after another on this
    GPU into one buffer must give exactly the bytes of the whole-frame render (same contributions, same order)."""
    path = onnx_model(model, scale, 2, tile, small=small)
    eng = make_engine(pkg, path, 2, tile, scale, tta=tta)
    frame = smooth_frame(*shape, 31)
    whole = eng.render(frame)
    for parts in (2, 3, 5):
        out = np.full_like(whole, 77)
        for part in range(parts):
            assert eng.render_strip(frame, out, part, parts), eng.last_error()
        assert np.array_equal(out, whole), parts
    # a strip writes nothing outside its own columns
    out = np.full_like(whole, 77)
    assert eng.render_strip(frame, out, 1, 2)
    _, cnt, x0, x1 = pkg.strip_plan(shape[1], shape[0], shape[1] * scale, shape[0] * scale, tile, eng.output_tile_size, scale, (0.0625, 0.0625), 1, 2)
    assert cnt > 0 and (out[:, :x0] == 77).all() and np.array_equal(out[:, x0:x1], whole[:, x0:x1])
    eng.close()


@pytest.mark.parametrize("shape,tta", [((200, 260), False), ((420, 560), False), ((101, 119), True)])
def test_rolling_sequence_of_frames_matches_render(pkg, onnx_model, monkeypatch, shape, tta):
    """renderSequence / benchResident on frames whose passes run as two tile groups ROLL (engine.cpp run_rolling_frame): no join at the end of a frame - the
    first stream goes on with the next frame's first group while the second stream composes the frame behind its own group, the tile slab alternating
    between two buffers.  Seven different frames (the fourth sighting of each buffer pair replays captured graphs), one pass per frame, two passes per
    frame (108 tiles of 64) and TTA (8 slots per tile): every frame is the bytes of render(), with output buffers reused as a ring, and W2X_NO_ROLLING=1
    gives the same; the engine is left usable and a resident replay leaves the last frame's output in place."""
    path = onnx_model("swin_unet/art", 4, 2, 64)
    frames = [smooth_frame(shape[0], shape[1], 60 + k) for k in range(7)]
    outs = {}
    for rolling in (True, False):
        if rolling: monkeypatch.delenv("W2X_NO_ROLLING", raising=False)
        else: monkeypatch.setenv("W2X_NO_ROLLING", "1")
        eng = make_engine(pkg, path, 2, 64, 4, tta=tta)
        want = [eng.render(f) for f in frames]
        host = [eng.alloc_host(f.shape) for f in frames]
        for h, f in zip(host, frames):
            h[...] = f
        got = eng.render_sequence(host, pinned=True)
        for k, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), f"frame {k}, rolling={rolling}"
        ring = [eng.alloc_host(want[0].shape) for _ in range(3)]
        for rep in range(2):                                                        # the second time every pass is a replayed graph
            eng.render_sequence(host, outs=[ring[k % 3] for k in range(7)])
            assert np.array_equal(ring[6 % 3], want[6]) and np.array_equal(ring[5 % 3], want[5]) and np.array_equal(ring[4 % 3], want[4])
        assert np.array_equal(eng.render(frames[2]), want[2])
        assert eng.bench_resident(5) > 0 and eng.bench_resident(4) > 0          # resident replays of the last frame roll too (odd and even counts)
        assert np.array_equal(eng.render(frames[3]), want[3])
        outs[rolling] = want
        eng.close()
    assert all(np.array_equal(a, b) for a, b in zip(outs[True], outs[False]))


def test_an_engine_loaded_again_rolls_and_replays_like_a_fresh_one(pkg, onnx_model):
    """load() on an engine that is already loaded releases everything the first load owned (img2img_load.cpp:149-154 resets engine and context the same way).
    Round 4's release() freed the second tile slab of a rolling sequence but kept its capacity, so after a re-load the rolling frames (render_sequence,
    bench_resident) swapped a null slab in: a device fault.  Here: load, render, roll, load again (other batch size: another plan), render, roll, replay -
    every frame is the bytes of a fresh engine's, and bench_resident() before any render() of the new load has no frame to replay."""
    path = onnx_model("swin_unet/art", 4, 2, 64)
    frames = [smooth_frame(200, 260, 90 + k) for k in range(4)]
    fresh = make_engine(pkg, path, 2, 64, 4)
    want = [fresh.render(f) for f in frames]
    fresh.close()
    eng = make_engine(pkg, path, 2, 64, 4)
    assert all(np.array_equal(a, b) for a, b in zip(eng.render_sequence(frames), want))
    assert eng.bench_resident(3) > 0
    for again in range(2):
        assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=4)), eng.last_error()
        assert eng.bench_resident(2) < 0                                       # nothing rendered since this load
        assert all(np.array_equal(a, b) for a, b in zip(eng.render_sequence(frames), want)), again
        assert np.array_equal(eng.render(frames[1]), want[1])
        assert eng.bench_resident(4) > 0 and eng.bench_resident(3) > 0
        assert np.array_equal(eng.render(frames[2]), want[2])
    eng.close()


@pytest.mark.parametrize("model,scale", [("swin_unet/art", 4), ("cunet/art", 2)])
def test_rewritten_graphs_render_the_bytes_of_the_original(pkg, onnx_model, tmp_path, model, scale):
    """Five re-spellings per family of the graph the exporter wrote (tools/onnx_rewrite.py: Gemm sandwiches, Identity / Cast / Transpose pairs, Constant nodes,
    0 / -1 Reshape targets, permuted node order ...; the CPU side is tests/test_loader_rewrites.py) built, loaded and rendered like any model file:
    the frames are the original's, byte for byte."""
    import onnx_rewrite as rw
    from oracle import onnx_reader
    path = onnx_model(model, scale, 2, 64, noise=1)
    frame = smooth_frame(150, 170, 21)
    eng = make_engine(pkg, path, 2, 64, scale)
    want = eng.render(frame)
    eng.close()
    g, shapes = onnx_reader.load(path), rw.runtime_shapes(path, 2, 64)
    for seed in range(5):
        v = rw.rewrite(g, shapes, 1000 + seed, kinds=rw.EXACT, count=5)
        vdir = tmp_path / f"v{seed}"; vdir.mkdir()
        vpath = str(vdir / os.path.basename(path))
        rw.dump(v, vpath, packed=bool(seed & 1))
        eng = make_engine(pkg, vpath, 2, 64, scale)
        got = eng.render(frame)
        eng.close()
        assert np.array_equal(got, want), (seed, v.applied)


@pytest.mark.parametrize("model,scale", [("swin_unet/art", 4), ("cunet/art", 2)])
def test_mutated_graphs_follow_the_oracle(pkg, onnx_model, tmp_path, model, scale):
    """Ten SEMANTIC mutants per family (tools/onnx_mutate.py: another LeakyRelu slope, Clip bounds, LayerNorm epsilon, attention scale, a transposed weight, another
    roll distance, a dropped residual or bias, a transposed bias table ... one change at one site each; the CPU side is tests/test_loader_mutations.py): whatever the
    loader does not refuse is built, run through w2x_infer and compared with the oracle executing the MUTANT, within the bounds of the unmutated graphs - and it must be
    far closer to the mutant than the mutant is to the original.  A lowering that fills in the value it expects instead of the one the file holds fails here."""
    import onnx_mutate as om
    import onnx_rewrite as rw
    from oracle import onnx_reader
    batch, tile = 2, 64
    path = onnx_model(model, scale, batch, tile, noise=1)
    g, shapes = onnx_reader.load(path), rw.runtime_shapes(path, batch, tile)
    rng = np.random.default_rng(5)
    x = rng.random((batch, 3, tile, tile), dtype=np.float32).astype(np.float16).astype(np.float32)
    y_orig = onnx_exec.Executor(path).run(x)
    ran, refused, kinds = 0, 0, []
    for seed in range(40):
        if ran >= 10:
            break
        v = om.mutate(g, shapes, seed)
        if v.applied[0] in kinds and len(kinds) < 6:          # one mutant per kind first
            continue
        vdir = tmp_path / f"m{seed}"; vdir.mkdir()
        vpath = str(vdir / os.path.basename(path))
        rw.dump(v, vpath, packed=bool(seed & 1))
        eng = pkg.Img2Img()
        if not eng.build(vpath, pkg.BuildConfig.fixed(batch, tile)):
            assert "cannot lower node" in eng.last_error() or "graph:" in eng.last_error() or "fold:" in eng.last_error(), (seed, v.applied, eng.last_error())
            refused += 1
            eng.close()
            continue
        assert eng.load(vpath, pkg.RenderConfig(batchSize=batch, height=tile, width=tile, scaling=scale)), eng.last_error()
        y = eng.infer(x)
        eng.close()
        ref32, ref16 = onnx_exec.Executor(vpath).run(x), oracle16(vpath)(x)
        moved = float(np.abs(ref32 - y_orig).max())
        r = network_report(f"mutant {model} s{scale} seed {seed}: {v.applied[0]} at {v.site}", y, ref16, ref32)
        assert r["max_ulp16"] <= NET_MAX_ULP16 and r["max_ulp16_vs_fp32_oracle"] <= NET_MAX_ULP16_VS_FP32 and r["mean_abs"] <= NET_MEAN_ABS, (seed, v.applied, v.site, r)
        if moved > 20 * ULP16:                                 # the mutation moved the output by far more than the engine's error: the engine is on the mutant's side
            assert float(np.abs(y - y_orig).max()) > 0.5 * moved, (seed, v.applied, v.site, moved)
        kinds.append(v.applied[0]); ran += 1
    print(f"{model}: {ran} mutants run ({sorted(set(kinds))}), {refused} refused")
    assert ran >= (10 if model.startswith("swin") else 6)


@pytest.mark.parametrize("pinned,small", [(False, True), (True, True), (True, False)])
def test_frame_sequence_with_overlapped_copies_matches_render(pkg, onnx_model, pinned, small):
    """renderSequence: upload / compute / download of consecutive frames overlap on three streams (two device buffers, events);
    every frame must come out exactly as from render(), also when output buffers are reused as a ring.  pinned: frame buffers
    from the engine's page-locked allocator (w2x_alloc_host) - the only case in which the copies really run beside the kernels."""
    path = onnx_model("swin_unet/art", 4, 2, 64, small=small)
    eng = make_engine(pkg, path, 2, 64, 4)
    frames = [smooth_frame(90, 130, 40 + k) for k in range(7)]
    want = [eng.render(f) for f in frames]
    if pinned:
        host = [eng.alloc_host(f.shape) for f in frames]
        for h, f in zip(host, frames):
            h[...] = f
        frames = host
    got = eng.render_sequence(frames, pinned=pinned)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    ring = [eng.alloc_host(want[0].shape) if pinned else np.empty_like(want[0]) for _ in range(3)]   # a writer that consumes frames in order
    eng.render_sequence(frames, outs=[ring[k % 3] for k in range(7)])
    assert np.array_equal(ring[6 % 3], want[6]) and np.array_equal(ring[5 % 3], want[5]) and np.array_equal(ring[4 % 3], want[4])
    assert np.array_equal(eng.render(frames[2]), want[2])                    # the engine is left in a usable state
    assert eng.last_render_ms > 0
    if pinned:
        # in-place page-locking takes whole pages only; a heap block is refused (and the call still works, unpinned)
        blk = np.zeros(5000, np.uint8)
        assert eng._L.w2x_pin_host(eng._h, blk.ctypes.data, blk.nbytes) == 0
        for h in frames + ring:
            eng.free_host(h)
    eng.close()


@pytest.mark.parametrize("small", [True, False])
def test_super_batching_is_bit_identical(pkg, onnx_model, monkeypatch, small):
    """One network pass may carry S reference batches (W2X_SUPERBATCH); frames, progress callbacks and infer() must not change."""
    path = onnx_model("swin_unet/art", 4, 2, 64, small=small)
    frame = smooth_frame(100, 140, 21)
    res = []
    for S in ("1", "3"):
        monkeypatch.setenv("W2X_SUPERBATCH", S)
        eng = make_engine(pkg, path, 2, 64, 4)
        assert eng.pass_tiles == 2 * int(S)
        prog = []
        eng.setProgressCallback(lambda c, t, s: prog.append((c, t)))
        out = eng.render(frame)
        x = np.random.default_rng(3).random((2, 3, 64, 64), dtype=np.float32)
        res.append((out, prog, eng.infer(x)))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("model,scale,batch,small,tta", [("swin_unet/art", 4, 1, False, False), ("swin_unet/art", 4, 4, False, True), ("cunet/art", 2, 2, False, False)])
def test_graph_replay_is_bit_identical_to_plain_launches(pkg, onnx_model, monkeypatch, model, scale, batch, small, tta):
    """A network pass is captured as a hipGraph the second time it is met and replayed afterwards (the reference's network is
    one enqueueV3, img2img_infer.cpp:80).  Same kernels, same arguments: first (plain), second (capture + launch) and later
    (replay) renders of a frame, render_sequence through two frame buffers, and an engine with W2X_NO_GRAPH must all give the
    same bytes; a frame of another size in between must not disturb the cached passes."""
    path = onnx_model(model, scale, batch, 64, small=small)
    frame, other = smooth_frame(100, 150, 5), smooth_frame(70, 90, 6)
    monkeypatch.setenv("W2X_NO_GRAPH", "1")
    eng = make_engine(pkg, path, batch, 64, scale, tta=tta)
    want, want_other = eng.render(frame), eng.render(other)
    eng.close()
    monkeypatch.delenv("W2X_NO_GRAPH")
    eng = make_engine(pkg, path, batch, 64, scale, tta=tta)
    for k in range(4):
        assert np.array_equal(eng.render(frame), want), k
        if k == 1:
            assert np.array_equal(eng.render(other), want_other)
    got = eng.render_sequence([frame, frame, frame, frame, frame], pinned=True)
    assert all(np.array_equal(g, want) for g in got)
    assert np.array_equal(eng.render(other), want_other) and np.array_equal(eng.render(other), want_other) and np.array_equal(eng.render(other), want_other)
    eng.close()


def test_engines_on_their_own_host_threads(pkg, onnx_model, monkeypatch):
    """one Img2Img per device, one host thread each.  Every public entry makes the engine's device current
    for its own duration, so engines can be driven from fresh threads whatever device those threads start on.
    On a one-GPU box W2X_DEVICE_MAP lets two logical devices share the physical one; with two GPUs they are distinct."""
    monkeypatch.setenv("W2X_DEVICE_MAP", "0,0")
    path = onnx_model("swin_unet/art", 4, 2, 64, small=False)
    frames = [smooth_frame(90, 130, 60 + k) for k in range(4)]
    ref_eng = make_engine(pkg, path, 2, 64, 4)
    want = [ref_eng.render(f) for f in frames]
    ref_eng.close()
    engs = []
    for dev in (0, 1):
        e = pkg.Img2Img()
        assert e.build(path, pkg.BuildConfig.fixed(2, 64, device=dev)), e.last_error()
        assert e.load(path, pkg.RenderConfig(deviceId=dev, batchSize=2, height=64, width=64, scaling=4)), e.last_error()
        engs.append(e)
    got = [None] * 4
    errs = []

    def work(e, idx):
        try:
            for i in idx:
                got[i] = e.render(frames[i])
            got[idx[0]] = e.render_sequence([frames[idx[0]]] * 3, pinned=True)[2]
            whole = np.zeros_like(got[idx[1]])
            for part in range(2):
                assert e.render_strip(frames[idx[1]], whole, part, 2)
            got[idx[1]] = whole
        except Exception as ex:      # surfaced in the main thread
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(engs[0], [0, 2])), threading.Thread(target=work, args=(engs[1], [1, 3]))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    [e.close() for e in engs]


@pytest.mark.parametrize("model,scale,tile,kernels", [
    ("cunet/art", 2, 96, 16),        # stem, 3x3 layers (k_conv3.hip, with and without pooling), both image heads (k_conv3h.hip), 1x1 + pixel shuffle
    ("cunet/art", 1, 64, 16),
    ("swin_unet/art", 4, 64, 6),     # stem, 48 -> 96 patch convolution, patch merges, pixel-shuffle projections, image head
])
def test_shape_specialised_kernels_agree_with_the_general_kernel(pkg, onnx_model, monkeypatch, model, scale, tile, kernels):
    """Every launch that a shape-specialised kernel takes (k_stem / k_conv3 / k_conv3h / k_conv48 / k_pixgemm) is repeated on
    gemm_kernel, the implicit-GEMM kernel that covers all of them (W2X_CHECK_GENERAL, engine.cpp): the two outputs may differ by
    the rounding of one fp16 value (different summation order; k_conv3h rounds once where gemm_kernel rounds before and after the
    skip add; the streaming projections round before their skip add, gemm_kernel after) - 2^-7 for activations in [8, 16), 2^-6 for the
    few in [16, 32) - and never by more."""
    path = onnx_model(model, scale, 2, tile, noise=1)
    monkeypatch.setenv("W2X_CHECK_GENERAL", "1")
    eng = pkg.Img2Img()
    lines = []
    eng.setMessageCallback(lambda sev, m: lines.append(m) if "pixgemm check" in m else None)
    assert eng.build(path, pkg.BuildConfig.fixed(2, tile)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=tile, width=tile, scaling=scale)), eng.last_error()
    monkeypatch.delenv("W2X_CHECK_GENERAL")
    eng.infer(np.random.default_rng(5).random((2, 3, tile, tile), dtype=np.float32))
    eng.close()
    import re
    seen = [(float(m.group(1)), int(m.group(2))) for m in (re.search(r"max\|d\|=([0-9.]+) .*?, (\d+) of \d+ off by", l) for l in lines) if m]
    assert len(seen) >= kernels, lines
    # (one fp16 value's rounding: 2^-7 for activations in [8, 16), 2^-6 in [16, 32) - the engine counts what is off by more than 0.01)
    worst = [l for l, (md, bad) in zip([l for l in lines if re.search(r"max\|d\|=([0-9.]+) .*?, (\d+) of \d+ off by", l)], seen) if bad > 2 or md > 2.0 ** -6]
    assert not worst, worst


FP32_NET_MAX_ABS = 2e-6   # Precision::FP32 against the fp32 oracle (outputs in [0, 1]): summation order only; measured <= 6.6e-7 (profiles/r2_final/parity.jsonl)
TF32_NET_MAX_ABS = 3e-5   # Precision::TF32 (split-bf16 products, 16 significant bits per operand): measured <= 8.9e-6, mean 1.4e-6 (profiles/r5_final/split_precision.txt);
                          # products on 11 significant bits - what TF32 keeps - would sit near 1e-3


@pytest.mark.parametrize("precision", ["FP32", "TF32"])
@pytest.mark.parametrize("model,scale,batch,tile", [("cunet/art", 2, 2, 64), ("cunet/art", 1, 1, 64), ("swin_unet/art", 4, 2, 64), ("swin_unet/photo", 2, 1, 88),
                                                    ("swin_unet/art_scan", 4, 1, 64)])
def test_fp32_engine_matches_the_fp32_oracle(pkg, onnx_model, model, scale, batch, tile, precision):
    """Precision::FP32 and Precision::TF32 build the fp32-storage engine (k_f32.hip: fp32 maps, un-fused operator set; exact fp32 MFMA products
    / three bf16 products per k-step).  Its output is compared with the fp32 oracle directly - no fp16 rounding on either side.  With FP32 what
    remains is summation order: this pins the lowering itself (LayerNorm folding, window tables and masks, pixel shuffles, crops,
    squeeze-excite) independently of the fp16 tolerances above, and frames agree with the oracle pipeline up to ties of rint().  With TF32
    the bound is the split's (above) and a few more pixels sit on the other side of a rounding boundary."""
    path = onnx_model(model, scale, batch, tile, noise=1)
    prec = pkg.Precision[precision]
    net_bound, frame_frac = (FP32_NET_MAX_ABS, 1e-3) if precision == "FP32" else (TF32_NET_MAX_ABS, 5e-3)
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(batch, tile, precision=prec)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=batch, height=tile, width=tile, scaling=scale, overlap=(0.0625, 0.0625))), eng.last_error()
    rng = np.random.default_rng(21)
    x = rng.random((batch, 3, tile, tile), dtype=np.float32)
    ex = onnx_exec.Executor(path)
    y, ref = eng.infer(x), ex.run(x)
    d = np.abs(y.astype(np.float64) - ref.astype(np.float64))
    from parity_util import _record
    _record({"test": f"network {precision.lower()} [{model} s{scale} B{batch} T{tile}]", "kind": "network_" + precision.lower(), "max_abs": float(d.max()), "mean_abs": float(d.mean())})
    assert d.max() <= net_bound, (d.max(), d.mean())
    frame = smooth_frame(tile + 37, 2 * tile - 9, 3) if scale > 1 else smooth_frame(70, 75, 3)   # (x1: 8-pixel output tiles, keep the grid small)
    out = eng.render(frame)
    want = pipeline.render(frame, ex.run, batch=batch, tile=tile, scaling=scale, overlap=(0.0625, 0.0625))
    r = frame_report(f"frame {precision.lower()} [{model} s{scale} B{batch} T{tile}]", out, want)
    assert r["max_lsb"] <= 1 and r["frac_pixels_off_by_1"] < frame_frac, r
    eng.close()
    if model == "cunet/art" and scale == 2:   # the fp32 tile path with TTA, as two strips
        eng = pkg.Img2Img()
        assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=batch, height=tile, width=tile, scaling=scale, overlap=(0.0625, 0.0625), tta=True)), eng.last_error()
        small = smooth_frame(70, 130, 5)
        want = pipeline.render(small, ex.run, batch=batch, tile=tile, scaling=scale, overlap=(0.0625, 0.0625), tta=True)
        out = np.zeros_like(want)
        for part in range(2):
            assert eng.render_strip(small, out, part, 2), eng.last_error()
        r = frame_report(f"frame {precision.lower()} tta strips [{model} s{scale} B{batch} T{tile}]", out, want)
        assert r["max_lsb"] <= 1 and r["frac_pixels_off_by_1"] < frame_frac, r
        eng.close()


def test_tf32_engine_at_the_headline_tile(pkg, onnx_model):
    """Precision::TF32 at the tile of configs[2] (T = 256, batch 2): 240 x 240 / 120 x 120 / 60 x 60 token maps, i.e. 1 600 / 400 / 100 windows per tile through
    swinattn32_kernel (three / two windows per workgroup, the last workgroup ragged at two tiles x 100 windows) and row counts that do not fill mlp32_kernel's last
    64-row workgroup at any level but the first.  Network only, against the fp32 oracle: measured 9.2e-6 (T = 400: 9.0e-6)."""
    path = onnx_model("swin_unet/art", 4, 2, 256)
    prec = pkg.Precision.TF32
    eng = pkg.Img2Img()
    assert eng.build(path, pkg.BuildConfig.fixed(2, 256, precision=prec)), eng.last_error()
    assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=2, height=256, width=256, scaling=4)), eng.last_error()
    x = np.random.default_rng(5).random((2, 3, 256, 256), dtype=np.float32)
    y = eng.infer(x)
    eng.close()
    d = np.abs(y.astype(np.float64) - onnx_exec.Executor(path).run(x))
    from parity_util import _record
    _record({"test": "network tf32 [swin_unet/art s4 B2 T256]", "kind": "network_tf32", "max_abs": float(d.max()), "mean_abs": float(d.mean())})
    assert d.max() <= TF32_NET_MAX_ABS, (d.max(), d.mean())


@pytest.mark.parametrize("model,scale,batch,tile", [("swin_unet/art", 4, 2, 64), ("swin_unet/photo", 2, 1, 88)])
def test_tf32_fused_launches_agree_with_the_unfused_plan(pkg, onnx_model, model, scale, batch, tile):
    """Precision::TF32 runs the fp32 plan's transformer blocks as two fused launches each (k_f32.hip: swinattn32_kernel for qkv gemm -> attention -> proj gemm,
    mlp32_kernel for fc1 -> fc2; round 6), Precision::FP32 and the debug switch no_fuse keep the un-fused launches.  Same split-bf16 products either way: the two
    TF32 engines agree to summation order (<= 2e-5 on outputs in [0, 1]; measured 1e-5), both stay inside the TF32 bound against the fp32 oracle, frames of a
    ragged size come out within one rounding tie of each other, and the fused engine reproduces itself bit for bit across batch groupings (two tile groups /
    one: the statistics' partial sums are added in a fixed order)."""
    path = onnx_model(model, scale, batch, tile, noise=1)
    prec = pkg.Precision.TF32
    x = np.random.default_rng(33).random((batch, 3, tile, tile), dtype=np.float32)
    ref = onnx_exec.Executor(path).run(x)
    frame = smooth_frame(tile + 51, 2 * tile + 13, 8)
    outs = []
    for nofuse in (0, 1):
        with pkg.debug_switches(no_fuse=nofuse):
            eng = pkg.Img2Img()
            assert eng.build(path, pkg.BuildConfig.fixed(batch, tile, precision=prec)), eng.last_error()
            assert eng.load(path, pkg.RenderConfig(precision=prec, batchSize=batch, height=tile, width=tile, scaling=scale, overlap=(0.0625, 0.0625))), eng.last_error()
            y = eng.infer(x)
            f1 = eng.render(frame)
            f2 = eng.render(frame)
            assert np.array_equal(f1, f2)
            outs.append((y, f1))
            eng.close()
    (y_f, fr_f), (y_u, fr_u) = outs
    from parity_util import _record
    d_fu = float(np.abs(y_f.astype(np.float64) - y_u).max())
    _record({"test": f"tf32 fused vs un-fused [{model} s{scale} B{batch} T{tile}]", "kind": "network_tf32_fusion", "max_abs_fused_vs_unfused": d_fu,
             "max_abs_fused_vs_oracle": float(np.abs(y_f - ref).max()), "max_abs_unfused_vs_oracle": float(np.abs(y_u - ref).max())})
    assert d_fu <= 2e-5 and np.abs(y_f - ref).max() <= TF32_NET_MAX_ABS and np.abs(y_u - ref).max() <= TF32_NET_MAX_ABS
    dfr = np.abs(fr_f.astype(np.int32) - fr_u.astype(np.int32))
    assert dfr.max() <= 1 and (dfr > 0).mean() < 5e-3


def test_folded_squeeze_excite_gates_equal_the_in_place_pass(pkg, onnx_model, monkeypatch):
    """cunet's squeeze-excite gates are folded into their consumers (the 1x1 / 2x2 (transposed) convolutions scale their operand
    on load, the skip add scales its residual: lower.cpp pending_gate) with the rounding of the separate in-place pass they
    replace (the debug switch no_se_fold keeps that pass): the network outputs and the frames are bit-identical."""
    path = onnx_model("cunet/art", 2, 2, 96, noise=1)
    x = np.random.default_rng(9).random((2, 3, 96, 96), dtype=np.float32)
    frame = smooth_frame(150, 170, 4)
    outs = []
    for nofold in (True, False):
        with pkg.debug_switches(no_se_fold=int(nofold)):
            assert (" scale t" in pkg.describe_plan(path, 2, 96)) == nofold
            eng = make_engine(pkg, path, 2, 96, 2)
        outs.append((eng.infer(x), eng.render(frame)))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("model,scale,tile,tta", [("cunet/art", 2, 64, False), ("swin_unet/art", 4, 64, True)])
def test_two_tile_groups_on_two_streams_are_bit_identical_to_one(pkg, onnx_model, monkeypatch, model, scale, tile, tta):
    """A pass is cut into two tile groups that run side by side on two streams, each in its own half of the activation arena
    (engine.cpp run_frame); W2X_GROUPS=1 keeps the pass in one piece.  Tiles never exchange data: same bytes, also through the
    captured graphs (three renders per engine) and for an odd number of live tiles."""
    path = onnx_model(model, scale, 2, tile, noise=1)
    frame = smooth_frame(150, 170, 8)
    outs = []
    for nosplit in (True, False):
        if nosplit: monkeypatch.setenv("W2X_GROUPS", "1")
        else: monkeypatch.delenv("W2X_GROUPS")
        eng = make_engine(pkg, path, 2, tile, scale, tta=tta)
        rs = [eng.render(frame) for _ in range(3)]
        assert np.array_equal(rs[0], rs[1]) and np.array_equal(rs[0], rs[2])
        outs.append(rs[0])
        eng.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("tile,batch,shape,tta", [(64, 2, (150, 170), False), (64, 2, (101, 119), True), (256, 4, (300, 420), False)])
def test_image_head_folded_into_the_last_mlp_is_bit_identical(pkg, onnx_model, monkeypatch, tile, batch, shape, tta):
    """The plan's last MLP and the image head behind it (Linear 96 -> 4x4 sub-pixels x 4 channels, Clip, DepthToSpace) run as ONE launch
    (engine.cpp fuse_head, k_mlp96q.hip): the 96-channel map between them is neither stored nor read back.  The debug switch no_fuse_head keeps the
    two launches.  Same sums in the same order, so infer() - whose output tensor lives in the activation arena, where the head's output
    must not be given the memory of the MLP's input - and render() - which writes into the frame slab - return the same bytes, also
    through captured graphs and two tile groups; the tile-256 engine makes every wave of the persistent MLP kernel walk several tiles."""
    path = onnx_model("swin_unet/art", 4, batch, tile, noise=3)
    frame = smooth_frame(shape[0], shape[1], 12)
    x = np.random.default_rng(31).random((batch, 3, tile, tile), dtype=np.float32)
    outs = []
    for nofuse in (True, False):
        with pkg.debug_switches(no_fuse_head=int(nofuse)):
            eng = make_engine(pkg, path, batch, tile, 4, tta=tta)
        ys = [eng.infer(x) for _ in range(2)]
        rs = [eng.render(frame) for _ in range(3)]
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(rs[0], rs[1]) and np.array_equal(rs[0], rs[2])
        outs.append((ys[0], rs[0]))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]), np.abs(outs[0][0] - outs[1][0]).max()
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("model,scale,tile,batch,shape,tta", [("swin_unet/art", 4, 64, 2, (150, 170), False), ("swin_unet/art", 4, 64, 3, (101, 119), True), ("swin_unet/art", 4, 256, 4, (300, 420), False),
                                                              ("cunet/art", 2, 64, 2, (150, 170), False), ("cunet/art", 2, 96, 3, (101, 119), True), ("cunet/art", 1, 64, 2, (90, 131), False),
                                                              ("cunet/art", 2, 256, 4, (300, 420), False), ("cunet/art", 2, 68, 1, (77, 131), False), ("cunet/art", 1, 100, 2, (120, 95), True)])
def test_stem_folded_into_the_patch_convolution_is_bit_identical(pkg, onnx_model, monkeypatch, model, scale, tile, batch, shape, tta):
    """swin_unet's first two ops - the stem (3x3, 4-halves-per-pixel tile -> 48 channels) and the patch convolution behind it - run as ONE launch
    (engine.cpp fuse_stem, k_conv48.hip conv48_kernel<true>): every workgroup computes the halo tile it needs from the input tile with the stem
    kernel's own instruction sequence, and the 48-channel map between the two is neither stored nor read.  The debug switch no_fuse_stem keeps the two
    launches.  Same products in the same order: infer() and render() return the same bytes, through captured graphs and two tile groups (the
    input tile must outlive the stem by one op in the arena); the odd batch leaves a group with one tile.
    Round 6: cunet's U-Nets open the same way (3x3 4 -> 32, then 3x3 32 -> 64: k_conv3.hip conv3_kernel<false, true> computes its one 32-channel chunk) - two
    stems folded at scale 2 (the second U-Net's input is the first one's output, which the residual at the end reads as well), one at scale 1."""
    path = onnx_model(model, scale, batch, tile, noise=1)
    frame = smooth_frame(shape[0], shape[1], 14)
    x = np.random.default_rng(37).random((batch, 3, tile, tile), dtype=np.float32)
    outs = []
    for nofuse in (True, False):
        with pkg.debug_switches(no_fuse_stem=int(nofuse)):
            eng = make_engine(pkg, path, batch, tile, scale, tta=tta)
        folded = [m for _, m in eng.messages if "stem folded" in m]
        assert bool(folded) == (not nofuse), [m for _, m in eng.messages if "folded" in m]
        if folded:
            assert f", {2 if model == 'cunet/art' else 1} stem folded" in folded[0], folded[0]
        ys = [eng.infer(x) for _ in range(2)]
        rs = [eng.render(frame) for _ in range(3)]
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(rs[0], rs[1]) and np.array_equal(rs[0], rs[2])
        outs.append((ys[0], rs[0]))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]), np.abs(outs[0][0] - outs[1][0]).max()
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("scale,tile,batch,shape,tta", [(2, 64, 2, (150, 170), False), (2, 96, 3, (101, 119), True), (1, 64, 2, (90, 131), False), (2, 256, 4, (300, 420), False),
                                                       (2, 68, 1, (77, 131), False), (1, 100, 2, (120, 95), True), (2, 132, 2, (160, 260), False)])
def test_transposed_convolution_folded_into_the_convolution_behind_it_is_bit_identical(pkg, onnx_model, scale, tile, batch, shape, tta):
    """cunet's decoders: ConvTranspose 2x2 stride 2 on the gated map, LeakyReLU, + the cropped skip map, then 3x3 64 -> 64.  As launches: a pixel-shuffle
    projection (k_pixgemm.hip) that writes the largest 64-channel map of the graph, and the convolution that reads it back.  engine.cpp fuse_up runs them as ONE
    launch (k_conv3.hip conv3_kernel UP): the convolution's halo fetch brings the skip pixels, each wave computes one sub-pixel class of the projection with
    pixgemm_kernel's products (operand roles exchanged) and adds it in LDS.  Same products, same roundings: infer() and render() return the same bytes as with
    the debug switch no_fuse_up, through captured graphs and tile groups; the tile sizes put the halo origin on both parities and leave ragged edge tiles."""
    path = onnx_model("cunet/art", scale, batch, tile, noise=1)
    frame = smooth_frame(shape[0], shape[1], 15)
    x = np.random.default_rng(41).random((batch, 3, tile, tile), dtype=np.float32)
    outs = []
    for nofuse in (True, False):
        with pkg.debug_switches(no_fuse_up=int(nofuse)):
            eng = make_engine(pkg, path, batch, tile, scale, tta=tta)
        folded = [m for _, m in eng.messages if "transposed convolution folded" in m]
        assert bool(folded) == (not nofuse), [m for _, m in eng.messages if "Loaded" in m]
        if folded:
            assert ", 2 transposed convolution folded" in folded[0], folded[0]     # the 64-channel pairs of both U-Nets (the 128-channel one keeps its launch)
        ys = [eng.infer(x) for _ in range(2)]
        rs = [eng.render(frame) for _ in range(3)]
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(rs[0], rs[1]) and np.array_equal(rs[0], rs[2])
        outs.append((ys[0], rs[0]))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]), (np.abs(outs[0][0] - outs[1][0]).max(), float((outs[0][0] != outs[1][0]).mean()))
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("scale,tile,batch,shape,tta", [(2, 64, 2, (150, 170), False), (2, 96, 3, (101, 119), True), (1, 64, 2, (90, 131), False), (2, 256, 4, (300, 420), False), (1, 112, 1, (200, 90), False),
                                                       (2, 68, 1, (77, 131), False), (1, 100, 2, (120, 95), True), (2, 132, 2, (160, 260), False)])
def test_image_heads_as_a_column_walk_are_bit_identical(pkg, onnx_model, scale, tile, batch, shape, tta):
    """cunet's two image heads (3x3 from 64 channels onto 3 channels + skip + clip, and onto 4 sub-pixels x 3 channels) are pure input streams.  k_conv3h.hip runs them as a
    walk down 64-column strips (conv3h_walk_kernel: a ring of eight input rows in LDS, four new rows requested under the products of the previous four, every row fetched
    once) instead of one halo tile per workgroup (conv3h_kernel, kept behind the debug switch no_conv3h_walk).  Same products in the same order, same epilogue: the same
    bytes from infer() and render(), on tiles whose maps are not multiples of 64 columns or of the row blocks, through captured graphs and tile groups."""
    path = onnx_model("cunet/art", scale, batch, tile, noise=1)
    frame = smooth_frame(shape[0], shape[1], 16)
    x = np.random.default_rng(43).random((batch, 3, tile, tile), dtype=np.float32)
    outs = []
    for nowalk in (True, False):
        with pkg.debug_switches(no_conv3h_walk=int(nowalk)):
            eng = make_engine(pkg, path, batch, tile, scale, tta=tta)
            ys = [eng.infer(x) for _ in range(2)]
            rs = [eng.render(frame) for _ in range(3)]
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(rs[0], rs[1]) and np.array_equal(rs[0], rs[2])
        outs.append((ys[0], rs[0]))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]), (np.abs(outs[0][0] - outs[1][0]).max(), float((outs[0][0] != outs[1][0]).mean()))
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name,kw,tile", [
    ("window 8 (64 tokens)", dict(variant={"ws": 8}), 80),
    ("4 / 8 heads of 24 / 48", dict(variant={"heads": 4}), 64),
    ("3 heads of 32 / 64", dict(variant={"heads": 3}), 64),
    ("static batch dimension", dict(dynamic=False), 64),
    ("opset 11", dict(opset=11), 64),
    ("opset 15", dict(opset=15), 64),
    # the attention traced in torchvision's shifted_window_attention operator order (tools/synth_models.py SwinBlock.attn_tv): a zero Pad to the window
    # multiple in front, the shift mask built in-graph from slice assignments and two masked_fill, whole-map Slices behind the reverse roll
    ("torchvision operator order", dict(variant={"tv": 1}), 64),
])
def test_loader_takes_graphs_it_was_not_written_around(pkg, onnx_model, name, kw, tile):
    """img2img_build.cpp:81-88 hands any ONNX file to the parser.  The fused kernels cover the release graphs' transformer shapes
    ((C, head) = (96, 16) / (192, 32), windows of 6 x 6); other shapes must still build and meet the un-fused path's bounds - windows
    of 8 x 8, other head counts and widths - and the same weights exported with a static batch dimension or at another opset must
    lower onto the same plan and give the same bytes as the opset-17 dynamic-batch file."""
    path = onnx_model("swin_unet/art", 4, 2, tile, noise=2, **kw)
    eng = make_engine(pkg, path, 2, tile, 4)
    assert any("ONNX graph" in m and "MatMul x" in m for _, m in eng.messages)        # build() lists what the parser was handed
    rng = np.random.default_rng(17)
    x = rng.random((2, 3, tile, tile), dtype=np.float32).astype(np.float16).astype(np.float32)
    y = eng.infer(x)
    eng.close()
    if "variant" in kw and "tv" not in kw["variant"]:
        r = network_report(f"network[swin_unet/art s4 B2 T{tile} {name}]", y, oracle16(path)(x), onnx_exec.Executor(path).run(x))
        assert "swinattn" not in pkg.describe_plan(path, 2, tile)                      # these shapes run on the general kernels
        assert r["max_ulp16"] <= NET_MAX_ULP16_UNFUSED and r["mean_abs"] <= NET_MEAN_ABS and r["max_ulp16_vs_fp32_oracle"] <= NET_MAX_ULP16_VS_FP32 + 0.5, r
    else:
        ref_path = onnx_model("swin_unet/art", 4, 2, tile, noise=2)
        ref = make_engine(pkg, ref_path, 2, tile, 4)
        assert np.array_equal(y, ref.infer(x)), name
        ref.close()
        assert "swinattn" in pkg.describe_plan(path, 2, tile)
