"""The reference's command line (src/main.cpp:18-153, 197-209, 240-257) rebuilt as waifu2x-tensorrt_amd/w2x: option set,
validation, model path / suffix / output naming, and the built-in still-image codecs.  No GPU needed: --print-config stops
after parsing, `convert` only touches the codecs."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W2X = os.path.join(ROOT, "waifu2x-tensorrt_amd", "w2x")
BASE = ["--model", "swin_unet/art", "--scale", "4", "--noise", "3", "--batchSize", "4", "--tileSize", "256"]


@pytest.fixture(scope="module", autouse=True)
def _built(pkg):          # pkg builds the library (and the CLI with it)
    assert os.path.exists(W2X), "w2x was not built"


def run(*args):
    return subprocess.run([W2X, *args], capture_output=True, text=True, timeout=120)


def test_render_options_and_derived_names(tmp_path):
    img = tmp_path / "picture.one.png"
    img.write_bytes(b"x")
    out = tmp_path / "o"; out.mkdir()
    r = run(*BASE, "render", "-i", str(img), "-o", str(out), "--tta", "--blend", "1/32", "--crf", "30", "--print-config")
    assert r.returncode == 0, r.stderr
    c = json.loads(r.stdout)
    assert c["command"] == "render" and c["tta"] and c["blend"] == 1 / 32 and c["crf"] == 30 and c["codec"] == "libx264" and c["pix_fmt"] == "yuv420p"
    assert c["model_path"] == "models/swin_unet/art/noise3_scale4x.onnx"                                  # main.cpp:201-204
    assert c["suffix"] == "(swin_unet_art)(noise3)(scale4)(tta)"                                          # main.cpp:205-209
    assert c["outputs"] == [str(out / "picture.one(swin_unet_art)(noise3)(scale4)(tta).png")]             # main.cpp:240-250
    # options may follow the subcommand (fallthrough), --nosuffix keeps the stem, defaults as in the reference
    r = run("render", "-i", str(img), "--nosuffix", "--model", "cunet/art", "--scale", "1", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "--print-config")
    c = json.loads(r.stdout)
    assert c["model_path"] == "models/cunet/art/noise0_.onnx" and c["suffix"] == "(cunet_art)(noise0)" and c["blend"] == 1 / 16 and c["device"] == 0
    assert c["outputs"] == [str(tmp_path / "picture.one.png")]
    r = run("--model", "swin_unet/photo", "--scale", "2", "--noise", "-1", "--batchSize", "2", "--tileSize", "400", "build", "--print-config")
    assert json.loads(r.stdout)["model_path"] == "models/swin_unet/photo/scale2x.onnx"
    for prec in ("fp16", "tf32", "fp32", "FP32"):      # main.cpp:76-84 has fp16 and tf32 (case-insensitive); fp32 = exact products on the same engine (include/w2x/config.h)
        r = run("--model", "cunet/art", "--scale", "2", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "--precision", prec, "build", "--print-config")
        assert r.returncode == 0 and json.loads(r.stdout)["precision"] == prec.lower(), (prec, r.stderr)


@pytest.mark.parametrize("args,msg", [
    (["--model", "cunet/art", "--scale", "4", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "build"], "cunet/art does not support scale factor 4."),
    (["--model", "cunet/art", "--scale", "1", "--noise", "-1", "--batchSize", "1", "--tileSize", "64", "build"], "Noise level -1 does not support scale factor 1."),
    (["--model", "esrgan", "--scale", "2", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "build"], "--model"),
    (["--model", "cunet/art", "--scale", "3", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "build"], "--scale"),
    (["--model", "cunet/art", "--scale", "2", "--noise", "0", "--batchSize", "0", "--tileSize", "64", "build"], "--batchSize"),
    (["--model", "cunet/art", "--scale", "2", "--noise", "0", "--batchSize", "1", "--tileSize", "100", "build"], "--tileSize"),
    (["--model", "cunet/art", "--scale", "2", "--noise", "0", "--batchSize", "1", "--tileSize", "64"], "subcommand"),
    (["--model", "cunet/art", "--scale", "2", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "build", "render"], "subcommand"),
    (["--scale", "2", "--noise", "0", "--batchSize", "1", "--tileSize", "64", "build"], "--model is required"),
    (BASE + ["render"], "--input is required"),
    (BASE + ["render", "-i", "/nonexistent/file.png"], "does not exist"),
    (BASE + ["render", "-i", ".", "--blend", "0.3"], "--blend"),
    (BASE + ["render", "-i", ".", "--crf", "52"], "--crf"),
    (BASE + ["--precision", "int8", "build"], "--precision"),
    (BASE + ["--precision", "bf16", "build"], "--precision"),
    (BASE + ["render", "-i", ".", "--tta", "--tta-mode", "median"], "--tta-mode"),
    (BASE + ["render", "-i", ".", "--tta-mode", "reference"], "needs --tta"),
    (BASE + ["build", "--bogus"], "not expected"),
])
def test_invalid_command_lines_are_rejected(args, msg):
    r = run(*args, "--print-config")
    assert r.returncode != 0 and msg in r.stderr, (r.returncode, r.stderr)


def test_tta_mode_selects_the_reference_accumulation(tmp_path):
    """main.cpp:118 has one --tta switch and img2img_render.cpp:313-316 one accumulation (not the mean).  The command line
    offers both: --tta alone averages, --tta-mode reference (or --tta-compat) asks for the reference's bytes (RenderConfig::ttaBugCompat)."""
    img = tmp_path / "a.png"; img.write_bytes(b"x")
    modes = {}
    for extra in ([], ["--tta-mode", "mean"], ["--tta-mode", "reference"], ["--tta-mode=Reference"], ["--tta-compat"]):
        r = run(*BASE, "render", "-i", str(img), "--tta", *extra, "--print-config")
        assert r.returncode == 0, r.stderr
        c = json.loads(r.stdout)
        assert c["tta"] and c["suffix"].endswith("(tta)")           # the output name is the reference's either way (main.cpp:205-209)
        modes[" ".join(extra)] = c["tta_mode"]
    assert modes == {"": "mean", "--tta-mode mean": "mean", "--tta-mode reference": "reference", "--tta-mode=Reference": "reference", "--tta-compat": "reference"}
    assert "--tta-mode" in run("--help").stdout


def test_help_lists_the_reference_flags():
    r = run("--help")
    assert r.returncode == 0
    for flag in ("--model", "--scale", "--noise", "--batchSize", "--tileSize", "--device", "--precision", "render", "build",
                 "--input", "--recursive", "--output", "--nosuffix", "--blend", "--tta", "--codec", "--pix_fmt", "--crf"):
        assert flag in r.stdout


def test_builtin_png_and_ppm_codecs_round_trip(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "rgb.png")
    assert run("convert", "-i", str(tmp_path / "rgb.png"), "-o", str(tmp_path / "a.ppm")).returncode == 0
    assert run("convert", "-i", str(tmp_path / "a.ppm"), "-o", str(tmp_path / "b.png")).returncode == 0
    assert np.array_equal(np.array(Image.open(tmp_path / "b.png")), a)
    variants = {"gray": Image.fromarray(a[..., 0]), "rgba": Image.fromarray(np.dstack([a, a[..., :1]])),
                "pal16": Image.fromarray(a).quantize(16), "pal256": Image.fromarray(a).quantize(256), "bilevel": Image.fromarray(a[..., 0] > 127)}
    for name, im in variants.items():       # gray, alpha (kept: written back as RGBA), 4/8-bit palette, 1-bit
        im.save(tmp_path / f"{name}.png")
        assert run("convert", "-i", str(tmp_path / f"{name}.png"), "-o", str(tmp_path / f"{name}_o.png")).returncode == 0, name
        assert np.array_equal(np.array(Image.open(tmp_path / f"{name}_o.png")), np.array(Image.open(tmp_path / f"{name}.png").convert("RGBA" if name == "rgba" else "RGB"))), name
    # 16-bit samples (gray, RGB, RGB + alpha; rows filtered with Sub so the two-byte pixel stride is exercised): the decoder keeps
    # the high byte, as cv::imread(IMREAD_COLOR) does for the reference (libpng strip_16)
    import struct, zlib
    def png16(path, arr16):                  # arr16: [h, w, ch] uint16, ch in {1, 3, 4}
        h, w, ch = arr16.shape
        ctype = {1: 0, 3: 2, 4: 6}[ch]
        be = arr16.astype(">u2").tobytes()
        stride, bpp = w * ch * 2, ch * 2
        raw = bytearray()
        for y in range(h):
            line = np.frombuffer(be[y * stride:(y + 1) * stride], np.uint8).astype(np.int32)
            left = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]])
            raw += b"\x01" + ((line - left) & 255).astype(np.uint8).tobytes()
        def chunk(t, body): return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
        open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    for ch in (1, 3, 4):
        a16 = rng.integers(0, 65536, (19, 23, ch), dtype=np.uint16)
        png16(tmp_path / f"deep{ch}.png", a16)
        assert run("convert", "-i", str(tmp_path / f"deep{ch}.png"), "-o", str(tmp_path / f"deep{ch}_o.png")).returncode == 0, ch
        want = (a16 >> 8).astype(np.uint8)
        want = np.repeat(want, 3, axis=2) if ch == 1 else want          # (16-bit RGBA keeps its alpha plane's high byte as well)
        assert np.array_equal(np.array(Image.open(tmp_path / f"deep{ch}_o.png")), want), ch
    # Adam7-interlaced files (PIL cannot write them: built here from the seven sub-images, filter None)
    def png_adam7(path, arr8):               # arr8: [h, w, 3] uint8
        h, w, _ = arr8.shape
        raw = bytearray()
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = arr8[y0::dy, x0::dx]
            if sub.size:
                for row in sub: raw += b"\x00" + row.tobytes()
        def chunk(t, body): return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
        open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    for shape in ((37, 53, 3), (3, 2, 3), (9, 1, 3)):
        ai = rng.integers(0, 256, shape, dtype=np.uint8)
        png_adam7(tmp_path / "ilace.png", ai)
        assert np.array_equal(np.array(Image.open(tmp_path / "ilace.png")), ai)          # the file is what PIL reads as the same image
        assert run("convert", "-i", str(tmp_path / "ilace.png"), "-o", str(tmp_path / "ilace_o.png")).returncode == 0, shape
        assert np.array_equal(np.array(Image.open(tmp_path / "ilace_o.png")), ai), shape
    r = run("convert", "-i", str(tmp_path / "missing.png"), "-o", str(tmp_path / "x.png"))
    assert r.returncode != 0 and "cannot open" in r.stderr


def _avi_frames(path):
    """Independent reader of an uncompressed 24-bit AVI (RIFF walk in Python) -> (frames [n, h, w, 3] BGR top-down, fps)."""
    import struct
    d = open(path, "rb").read()
    assert d[:4] == b"RIFF" and d[8:12] == b"AVI "
    assert struct.unpack_from("<I", d, 4)[0] == len(d) - 8
    w = h = None; frames = []; fps = None
    def walk(pos, end):
        nonlocal w, h, fps
        while pos + 8 <= end:
            cid, n = d[pos:pos + 4], struct.unpack_from("<I", d, pos + 4)[0]
            body = pos + 8
            if cid == b"LIST":
                walk(body + 4, body + n)
            elif cid == b"strh":
                scale, rate = struct.unpack_from("<II", d, body + 20); fps = rate / scale
            elif cid == b"strf":
                w, h, _, bits, comp = struct.unpack_from("<iiHHI", d, body + 4); assert bits == 24 and comp == 0 and h > 0
            elif cid == b"00db":
                stride = (w * 3 + 3) // 4 * 4
                rows = np.frombuffer(d, np.uint8, stride * h, body).reshape(h, stride)[::-1, :w * 3]
                frames.append(rows.reshape(h, w, 3).copy())
            elif cid == b"idx1":
                assert n == 16 * len(frames)
            pos = body + (n + 1) // 2 * 2
    walk(12, len(d))
    return np.stack(frames), fps


def test_builtin_bmp_and_uncompressed_avi_codecs(tmp_path):
    """BMP (24-bit, 32-bit with alpha, odd widths: rows padded to 4 bytes, bottom-up) and the uncompressed AVI container through
    `w2x convert` (stills) and the codec classes as the CLI uses them; PIL reads what the writer wrote and the reader reads what PIL wrote."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    for shape in ((37, 53), (5, 1), (8, 2), (3, 7)):
        a = rng.integers(0, 256, (*shape, 3), dtype=np.uint8)
        Image.fromarray(a).save(tmp_path / "p.bmp")                                   # PIL's writer -> our reader
        assert run("convert", "-i", str(tmp_path / "p.bmp"), "-o", str(tmp_path / "p.png")).returncode == 0
        assert np.array_equal(np.array(Image.open(tmp_path / "p.png")), a), shape
        assert run("convert", "-i", str(tmp_path / "p.png"), "-o", str(tmp_path / "q.bmp")).returncode == 0   # our writer -> PIL's reader
        assert np.array_equal(np.array(Image.open(tmp_path / "q.bmp").convert("RGB")), a), shape
    rgba = rng.integers(0, 256, (21, 30, 4), dtype=np.uint8)
    Image.fromarray(rgba).save(tmp_path / "a.png")
    assert run("convert", "-i", str(tmp_path / "a.png"), "-o", str(tmp_path / "a.bmp")).returncode == 0      # 32-bit BGRA
    assert run("convert", "-i", str(tmp_path / "a.bmp"), "-o", str(tmp_path / "a2.png")).returncode == 0
    assert np.array_equal(np.array(Image.open(tmp_path / "a2.png")), rgba)
    r = run("convert", "-i", str(tmp_path / "a.png"), "-o", str(tmp_path / "clip.avi"))
    assert r.returncode != 0                                                           # a still is not a video container


@pytest.mark.gpu
def test_cli_build_and_render_match_the_library(pkg, tmp_path):
    """End to end on the GPU box: `w2x build` then `w2x render` on a PNG give the bytes Img2Img.render gives."""
    Image = pytest.importorskip("PIL.Image")
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "swin_unet/art", 4, 3)          # <tmp>/models/swin_unet/art/noise3_scale4x.onnx
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=5, small=True), path, 2, 64, dynamic=True)
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "in.png")
    out = tmp_path / "out"; out.mkdir()
    common = ["--models", str(models), "--model", "swin_unet/art", "--scale", "4", "--noise", "3", "--batchSize", "2", "--tileSize", "64"]
    r = run(*common, "build")
    assert r.returncode == 0, r.stderr
    r = run(*common, "render", "-i", str(tmp_path / "in.png"), "-o", str(out))
    assert r.returncode == 0, r.stderr
    assert "batch" in r.stderr                                          # progress line, main.cpp:190-193
    got = np.array(Image.open(out / "in(swin_unet_art)(noise3)(scale4).png"))
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=4)), eng.last_error()
    ref = eng.render(np.ascontiguousarray(rgb[..., ::-1]))              # the library works on BGR
    eng.close()
    assert np.array_equal(got, ref[..., ::-1])


@pytest.mark.gpu
def test_cli_two_devices_render_a_still_as_strips_and_a_tf32_engine(pkg, tmp_path):
    """`--devices N` on a still: one engine per device.  --split shards (default): every tile once, seam bands exchanged, each engine composes
    its cells of the shared output image (Img2Img::renderSharded); --split strips: each renders its tile-column strip, seam column recomputed
    (SURVEY 8e); the bytes are those of the single-device frame either way.  The box has one GPU, so W2X_DEVICE_MAP=0,0 puts both logical
    devices on it (engine.cpp physical_device).  The same command line with --precision tf32 builds and renders on the fp32 engine."""
    Image = pytest.importorskip("PIL.Image")
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "cunet/art", 2, 1)
    sm.export_onnx(sm.make_model("cunet/art", 2, seed=6), path, 2, 64, dynamic=True)
    rgb = np.random.default_rng(2).integers(0, 256, (150, 210, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "in.png")
    common = ["--models", str(models), "--model", "cunet/art", "--scale", "2", "--noise", "1", "--batchSize", "2", "--tileSize", "64"]
    env = dict(os.environ, W2X_DEVICE_MAP="0,0,0")
    def cli(*a):
        return subprocess.run([W2X, *common, *a], capture_output=True, text=True, timeout=180, env=env)
    r = cli("build")
    assert r.returncode == 0, r.stderr
    outs = {}
    for name, extra in (("one", []), ("two", ["--devices", "2"]), ("two_strips", ["--devices", "2", "--split", "strips"]), ("three", ["--devices", "3"])):
        out = tmp_path / name; out.mkdir()
        r = cli("render", "-i", str(tmp_path / "in.png"), "-o", str(out), *extra)
        assert r.returncode == 0, r.stderr
        outs[name] = np.array(Image.open(out / "in(cunet_art)(noise1)(scale2).png"))
    assert outs["one"].shape == (300, 420, 3)
    for name in ("two", "two_strips", "three"):
        assert np.array_equal(outs["one"], outs[name]), name
    r = cli("--precision", "tf32", "build")
    assert r.returncode == 0, r.stderr
    out = tmp_path / "tf32"; out.mkdir()
    r = cli("--precision", "tf32", "render", "-i", str(tmp_path / "in.png"), "-o", str(out))
    assert r.returncode == 0, r.stderr
    got32 = np.array(Image.open(out / "in(cunet_art)(noise1)(scale2).png")).astype(np.int32)
    assert np.abs(got32 - outs["one"].astype(np.int32)).max() <= 2       # fp16 against fp32 engine: a couple of LSBs at most
    # --precision fp32 (not in the reference: exact products on the same engine) builds its own engine file and renders within rounding ties of the tf32 frame
    r = cli("--precision", "fp32", "build")
    assert r.returncode == 0, r.stderr
    assert len([f for f in os.listdir(os.path.dirname(path)) if f.endswith(".w2x")]) >= 3, os.listdir(os.path.dirname(path))
    out = tmp_path / "fp32"; out.mkdir()
    r = cli("--precision", "fp32", "render", "-i", str(tmp_path / "in.png"), "-o", str(out))
    assert r.returncode == 0, r.stderr
    gotx = np.array(Image.open(out / "in(cunet_art)(noise1)(scale2).png")).astype(np.int32)
    assert np.abs(gotx - got32).max() <= 1 and (gotx != got32).mean() < 5e-3


FAKE_FFPROBE = """#!/usr/bin/env python3
# stand-in for ffprobe on a raw bgr24 clip: prints width,height,r_frame_rate,nb_read_packets like `-of csv=p=0`
import os, sys
w, h = int(os.environ["FAKE_W"]), int(os.environ["FAKE_H"])
print(f"{w},{h},30/1,{os.path.getsize(sys.argv[-1]) // (w * h * 3)}")
"""

FAKE_FFMPEG = """#!/usr/bin/env python3
# stand-in for ffmpeg: `-i FILE ... -` copies the raw clip to stdout, `-i - ... OUT` copies stdin to OUT
import shutil, sys
a = sys.argv[1:]
src = a[a.index("-i") + 1]
if src == "-":
    with open(a[-1], "wb") as f: shutil.copyfileobj(sys.stdin.buffer, f)
else:
    with open(src, "rb") as f: shutil.copyfileobj(f, sys.stdout.buffer)
"""


@pytest.mark.gpu
def test_cli_video_path_matches_the_library(pkg, tmp_path):
    """The video path (frames piped through ffmpeg as raw bgr24, reader / renderer / writer on their own threads) writes,
    in order, the frames Img2Img.render gives.  ffmpeg is not in the image: two small scripts stand in for the pipes."""
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "swin_unet/art", 4, 3)
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=5, small=True), path, 2, 64, dynamic=True)
    W, H, N = 100, 70, 10                                               # 10 frames: two full chunks of 4 and a ragged one
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)         # BGR
    (tmp_path / "clip.mp4").write_bytes(frames.tobytes())
    bindir = tmp_path / "bin"; bindir.mkdir()
    for name, text in (("ffprobe", FAKE_FFPROBE), ("ffmpeg", FAKE_FFMPEG)):
        (bindir / name).write_text(text); (bindir / name).chmod(0o755)
    out = tmp_path / "out"; out.mkdir()
    common = ["--models", str(models), "--model", "swin_unet/art", "--scale", "4", "--noise", "3", "--batchSize", "2", "--tileSize", "64"]
    env = dict(os.environ, PATH=f"{bindir}:{os.environ['PATH']}", FAKE_W=str(W), FAKE_H=str(H))
    r = subprocess.run([W2X, *common, "build"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([W2X, *common, "render", "-i", str(tmp_path / "clip.mp4"), "-o", str(out)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = (out / "clip(swin_unet_art)(noise3)(scale4).mp4").read_bytes()
    got = np.frombuffer(raw, np.uint8).reshape(N, H * 4, W * 4, 3)
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=4)), eng.last_error()
    for k in range(N):
        assert np.array_equal(got[k], eng.render(np.ascontiguousarray(frames[k]))), k
    eng.close()


def _write_avi(path, frames, fps=25):
    """Uncompressed 24-bit AVI the way `ffmpeg -c:v rawvideo -pix_fmt bgr24 x.avi` lays it out (bottom-up DIB rows, JUNK padding before
    movi, an audio-less single stream), written independently of the CLI's own writer."""
    import struct
    n, h, w, _ = frames.shape
    stride = (w * 3 + 3) // 4 * 4
    def chunk(cid, body): return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")
    def lst(t, body): return b"LIST" + struct.pack("<I", len(body) + 4) + t + body
    avih = struct.pack("<14I", int(1e6 / fps), 0, 0, 0x10, n, 0, 1, stride * h, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"\0\0\0\0" + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1, fps, 0, n, stride * h, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, stride * h, 0, 0, 0, 0)
    movi = b""
    for f in frames:
        rows = np.zeros((h, stride), np.uint8); rows[:, :w * 3] = f.reshape(h, w * 3)
        movi += chunk(b"00db", rows[::-1].tobytes())
    body = b"AVI " + lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf))) + chunk(b"JUNK", b"\0" * 12) + lst(b"movi", movi)
    open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [1, 2, 3])
def test_cli_video_loop_on_uncompressed_avi_matches_the_library(pkg, tmp_path, devices):
    """The frame loop of main.cpp:263-269 end to end WITHOUT ffmpeg and without stand-in scripts: an uncompressed AVI in (built-in reader),
    chunks of four frames round-robin over 1 / 2 / 3 engines (one persistent renderSequence worker each, page-locked slots; the box has
    one GPU, W2X_DEVICE_MAP puts the logical devices on it), an in-order writer, an uncompressed AVI out.  11 frames = two full chunks
    and a ragged one, so with 2 and 3 engines the last chunks finish out of order and must still be written in order: every output
    frame equals Img2Img.render of its input frame, byte for byte."""
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "swin_unet/art", 4, 3)
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=5, small=True), path, 2, 64, dynamic=True)
    W, H, N = 99, 70, 11                                                # odd width: padded DIB rows on both sides
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)         # BGR
    _write_avi(tmp_path / "clip.avi", frames, fps=24)
    out = tmp_path / "out"; out.mkdir()
    common = ["--models", str(models), "--model", "swin_unet/art", "--scale", "4", "--noise", "3", "--batchSize", "2", "--tileSize", "64"]
    env = dict(os.environ, W2X_DEVICE_MAP=",".join(["0"] * devices))
    env["PATH"] = os.pathsep.join(p for p in env["PATH"].split(os.pathsep) if not os.path.exists(os.path.join(p, "ffmpeg")))   # the built-in path even where ffmpeg exists
    r = subprocess.run([W2X, *common, "build"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([W2X, *common, "render", "-i", str(tmp_path / "clip.avi"), "-o", str(out), "--devices", str(devices)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "writing uncompressed video" in r.stderr
    got, fps = _avi_frames(out / "clip(swin_unet_art)(noise3)(scale4).avi")
    assert got.shape == (N, H * 4, W * 4, 3) and fps == 24
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=4)), eng.last_error()
    for k in range(N):
        assert np.array_equal(got[k], eng.render(np.ascontiguousarray(frames[k]))), k
    eng.close()


@pytest.mark.gpu
def test_cli_keeps_the_alpha_channel_of_a_still(pkg, tmp_path):
    """README.md:88 lists alpha as a TODO upstream.  Here an RGBA PNG comes out as an RGBA PNG: colour through the engine as usual, the
    alpha plane through the same engine as a gray image (its green channel is the new alpha)."""
    Image = pytest.importorskip("PIL.Image")
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "cunet/art", 2, 1)
    sm.export_onnx(sm.make_model("cunet/art", 2, seed=6), path, 2, 64, dynamic=True)
    rng = np.random.default_rng(5)
    rgba = rng.integers(0, 256, (80, 100, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:80, 0:100]
    rgba[..., 3] = np.clip(255 - np.hypot(yy - 40, xx - 50) * 5, 0, 255).astype(np.uint8)     # a soft disc
    Image.fromarray(rgba).save(tmp_path / "in.png")
    common = ["--models", str(models), "--model", "cunet/art", "--scale", "2", "--noise", "1", "--batchSize", "2", "--tileSize", "64"]
    assert subprocess.run([W2X, *common, "build"], capture_output=True, text=True).returncode == 0
    out = tmp_path / "o"; out.mkdir()
    r = subprocess.run([W2X, *common, "render", "-i", str(tmp_path / "in.png"), "-o", str(out)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    got = np.array(Image.open(out / "in(cunet_art)(noise1)(scale2).png"))
    assert got.shape == (160, 200, 4)
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=2)), eng.last_error()
    colour = eng.render(np.ascontiguousarray(rgba[..., 2::-1]))
    alpha = eng.render(np.ascontiguousarray(np.repeat(rgba[..., 3:4], 3, axis=2)))
    eng.close()
    assert np.array_equal(got[..., :3], colour[..., ::-1]) and np.array_equal(got[..., 3], alpha[..., 1])


@pytest.mark.gpu
def test_cli_deep_keeps_sixteen_bit_pngs(pkg, tmp_path):
    """--deep (extension): a 16-bit PNG is read with all 16 bits, rendered as CV_16UC3 and written as a 16-bit PNG - the samples the library's
    16-bit render() gives; without the flag the same file is cut to 8 bits like cv::imread(IMREAD_COLOR) does for the reference."""
    Image = pytest.importorskip("PIL.Image")
    import struct, zlib
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "cunet/art", 2, 1)
    sm.export_onnx(sm.make_model("cunet/art", 2, seed=6), path, 2, 64, dynamic=True)
    rng = np.random.default_rng(8)
    a16 = rng.integers(0, 65536, (50, 66, 3), dtype=np.uint16)              # RGB
    def chunk(t, body): return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    raw = b"".join(b"\x00" + a16[y].astype(">u2").tobytes() for y in range(a16.shape[0]))
    (tmp_path / "in.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 66, 50, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    common = ["--models", str(models), "--model", "cunet/art", "--scale", "2", "--noise", "1", "--batchSize", "2", "--tileSize", "64"]
    assert subprocess.run([W2X, *common, "build"], capture_output=True, text=True).returncode == 0
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=2)), eng.last_error()
    want16 = eng.render(np.ascontiguousarray(a16[..., ::-1]))[..., ::-1]
    want8 = eng.render(np.ascontiguousarray((a16 >> 8).astype(np.uint8)[..., ::-1]))[..., ::-1]
    eng.close()
    for flag, want in ((["--deep"], want16), ([], want8)):
        out = tmp_path / ("o16" if flag else "o8"); out.mkdir()
        r = subprocess.run([W2X, *common, "render", "-i", str(tmp_path / "in.png"), "-o", str(out), *flag], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        d = open(out / "in(cunet_art)(noise1)(scale2).png", "rb").read()
        w, h, depth, ctype = struct.unpack(">IIBB", d[16:26])
        assert (w, h, depth, ctype) == (132, 100, 16 if flag else 8, 2)
        idat = b"".join(d[p + 8:p + 8 + struct.unpack(">I", d[p:p + 4])[0]] for p in _png_chunks(d) if d[p + 4:p + 8] == b"IDAT")
        rows = zlib.decompress(idat)
        bps = 2 if flag else 1
        stride = 1 + w * 3 * bps
        got = np.stack([np.frombuffer(rows[y * stride + 1:(y + 1) * stride], ">u2" if flag else np.uint8).reshape(w, 3) for y in range(h)])   # filter 0 rows
        assert np.array_equal(got.astype(want.dtype), want)


def _png16(path, a16):
    """a16: (h, w, 3 or 4) uint16 RGB(A) -> 16-bit PNG, filter 0 rows"""
    import struct, zlib
    def chunk(t, body): return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    h, w, ch = a16.shape
    raw = b"".join(b"\x00" + a16[y].astype(">u2").tobytes() for y in range(h))
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2 if ch == 3 else 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def _read_png_rows(path):
    import struct, zlib
    d = open(path, "rb").read()
    w, h, depth, ctype = struct.unpack(">IIBB", d[16:26])
    idat = b"".join(d[p + 8:p + 8 + struct.unpack(">I", d[p:p + 4])[0]] for p in _png_chunks(d) if d[p + 4:p + 8] == b"IDAT")
    rows = zlib.decompress(idat)
    ch = {2: 3, 6: 4}[ctype]
    stride = 1 + w * ch * (depth // 8)
    assert all(rows[y * stride] == 0 for y in range(h))                       # the built-in writer emits filter 0 rows
    return np.stack([np.frombuffer(rows[y * stride + 1:(y + 1) * stride], ">u2" if depth == 16 else np.uint8).reshape(w, ch) for y in range(h)]), depth


@pytest.mark.gpu
def test_cli_deep_on_a_sixteen_bit_rgba_png(pkg, tmp_path):
    """--deep on colour types 4 / 6: the colour planes travel as CV_16UC3 (Bitmap::bgr16, Bitmap::bgr stays empty), the alpha plane's high
    byte goes through the same engine as an 8-bit gray image and comes back widened (x257).  The alpha scratch images were once sized
    from the empty 8-bit plane (heap overflow): the sizes now come from the geometry."""
    import synth_models as sm
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "cunet/art", 2, 1)
    sm.export_onnx(sm.make_model("cunet/art", 2, seed=6), path, 2, 64, dynamic=True)
    rng = np.random.default_rng(18)
    a16 = rng.integers(0, 65536, (41, 59, 4), dtype=np.uint16)               # RGBA
    yy, xx = np.mgrid[0:41, 0:59]
    a16[..., 3] = (np.clip(255 - np.hypot(yy - 20, xx - 30) * 9, 0, 255).astype(np.uint16) * 257)
    _png16(tmp_path / "in.png", a16)
    common = ["--models", str(models), "--model", "cunet/art", "--scale", "2", "--noise", "1", "--batchSize", "2", "--tileSize", "64"]
    assert subprocess.run([W2X, *common, "build"], capture_output=True, text=True).returncode == 0
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=2)), eng.last_error()
    colour16 = eng.render(np.ascontiguousarray(a16[..., 2::-1]))[..., ::-1]
    colour8 = eng.render(np.ascontiguousarray((a16[..., 2::-1] >> 8).astype(np.uint8)))[..., ::-1]
    alpha8 = eng.render(np.ascontiguousarray(np.repeat((a16[..., 3:4] >> 8).astype(np.uint8), 3, axis=2)))[..., 1]
    eng.close()
    for flag, colour in ((["--deep"], colour16), ([], colour8)):
        out = tmp_path / ("o16" if flag else "o8"); out.mkdir()
        r = subprocess.run([W2X, *common, "render", "-i", str(tmp_path / "in.png"), "-o", str(out), *flag], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        got, depth = _read_png_rows(out / "in(cunet_art)(noise1)(scale2).png")
        assert depth == (16 if flag else 8) and got.shape == (82, 118, 4)
        assert np.array_equal(got[..., :3].astype(colour.dtype), colour)
        assert np.array_equal(got[..., 3].astype(np.int64), alpha8.astype(np.int64) * (257 if flag else 1))


@pytest.mark.gpu
def test_cli_tta_reference_mode_gives_the_reference_accumulation(pkg, tmp_path):
    """`--tta --tta-mode reference` == the oracle pipeline with tta_bug_compat (img2img_render.cpp:305-318 as written, Q1) around the engine's
    own network, byte for byte; plain `--tta` == the oracle's true mean; and the two differ."""
    Image = pytest.importorskip("PIL.Image")
    import synth_models as sm
    from oracle import pipeline
    models = tmp_path / "models"
    path = sm.model_path(str(tmp_path), "swin_unet/art", 4, 3)
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=5, small=True), path, 2, 64, dynamic=True)
    rng = np.random.default_rng(21)
    rgb = rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "in.png")
    common = ["--models", str(models), "--model", "swin_unet/art", "--scale", "4", "--noise", "3", "--batchSize", "2", "--tileSize", "64"]
    assert run(*common, "build").returncode == 0
    eng = pkg.Img2Img()
    assert eng.load(path, pkg.RenderConfig(batchSize=2, height=64, width=64, scaling=4, tta=True)), eng.last_error()
    bgr = np.ascontiguousarray(rgb[..., ::-1])
    outs = {}
    for mode, bug in (("mean", False), ("reference", True)):
        out = tmp_path / mode; out.mkdir()
        r = run(*common, "render", "-i", str(tmp_path / "in.png"), "-o", str(out), "--tta", "--tta-mode", mode)
        assert r.returncode == 0, r.stderr
        got = np.array(Image.open(out / "in(swin_unet_art)(noise3)(scale4)(tta).png"))[..., ::-1]
        ref = pipeline.render(bgr, eng.infer, batch=2, tile=64, scaling=4, overlap=(0.0625, 0.0625), tta=True, tta_bug_compat=bug,
                              net_dtype=np.float16, tile_out=eng.output_tile_size)
        assert np.array_equal(got, ref), mode
        outs[mode] = got
    eng.close()
    assert not np.array_equal(outs["mean"], outs["reference"])


def _png_chunks(d):
    import struct
    p = 8
    while p + 12 <= len(d):
        yield p
        p += 12 + struct.unpack(">I", d[p:p + 4])[0]
