"""Everything in the tree that parses untrusted files, under AddressSanitizer + UBSan on the CPU (the reference has
no sanitizer target, CMakeLists.txt:49-68): the hand-written ONNX protobuf reader, constant folding and lowering (what build() runs on
a model file, img2img_build.cpp:81-88), the engine-file reader (what load() runs, img2img_load.cpp:149-154), the built-in PNG / BMP /
PPM / AVI codecs of the CLI and its option parser.

`make asan` builds waifu2x-tensorrt_amd/w2x_parse_check (csrc/cli/parse_check.cpp: host sources only, g++ -fsanitize=address,undefined
-fno-sanitize-recover=all).  Each case feeds it one valid, truncated or bit-flipped file (seeded corpus, generated here) and accepts two
outcomes: exit 0 (parsed) or exit 2 (rejected with a message - the product's clean `false`).  A crash, a sanitizer report, an
allocation bomb or a hang fails the test.  The first run of this target found a use-after-free in the squeeze-excite lowering of
every VALID cunet graph (lower.cpp, a TensorDesc reference held across a vector growth)."""
import os
import struct
import subprocess
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "waifu2x-tensorrt_amd")
CHECK = os.path.join(PKG, "w2x_parse_check")
ENV = dict(os.environ, ASAN_OPTIONS="exitcode=97:detect_leaks=1:allocator_may_return_null=0:max_allocation_size_mb=2048", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module", autouse=True)
def _built():
    r = subprocess.run(["make", "-C", PKG, "asan"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(CHECK), r.stderr[-2000:]


def check(*args, timeout=120):
    r = subprocess.run([CHECK, *map(str, args)], capture_output=True, timeout=timeout, env=ENV)
    return r.returncode, (r.stdout + r.stderr).decode("utf-8", "replace")      # messages quote names from the (mutated) file


def run_corpus(mode, files, extra=(), workers=6):
    """every file through the driver; returns {name: rc}; fails on anything but accept (0) / reject (2)"""
    def one(item):
        name, path = item
        rc, out = check(mode, path, *extra)
        return name, rc, out
    with ThreadPoolExecutor(workers) as ex:
        res = list(ex.map(one, files.items()))
    bad = [(n, rc, out[-1500:]) for n, rc, out in res if rc not in (0, 2)]
    assert not bad, f"{len(bad)} of {len(res)} inputs crashed the {mode} parser; first: {bad[0]}"
    for n, rc, out in res:
        if rc == 2:
            assert "rejected: " in out and len(out.strip().splitlines()[-1]) > len("rejected: "), (n, out)      # a message, not a bare failure
    return {n: rc for n, rc, _ in res}


def mutants(data, rng, n_trunc, n_flip, region=None, tag=""):
    """seeded truncations and single-byte changes (random value, 0x00, 0xFF, high bit) of `data`; region = (lo, hi) limits the flips"""
    out = {}
    L = len(data)
    cuts = sorted({0, 1, 7, 8, 33, L // 4, L // 2, L - 1001, L - 9, L - 1} | {int(v) for v in rng.integers(0, L, max(n_trunc - 10, 0))})
    for c in cuts[:max(n_trunc, 0)] if n_trunc < len(cuts) else cuts:
        if 0 <= c < L:
            out[f"{tag}cut{c}"] = data[:c]
    lo, hi = region or (0, L)
    for k in range(n_flip):
        pos = int(rng.integers(lo, min(hi, L)))
        b = bytearray(data)
        b[pos] = (int(rng.integers(0, 256)), 0x00, 0xFF, b[pos] ^ 0x80)[k % 4]
        out[f"{tag}flip{pos}_{k % 4}"] = bytes(b)
    return out


def write_all(tmp_path, blobs, ext):
    files = {}
    for name, data in blobs.items():
        p = tmp_path / f"{name}{ext}"
        p.write_bytes(data)
        files[name] = str(p)
    return files


# ----------------------------------------------------------------------------------------------------------------- ONNX + engine files
@pytest.fixture(scope="module")
def small_models(tmp_path_factory):
    import synth_models as sm
    root = tmp_path_factory.mktemp("malformed_models")
    out = {}
    for model, scale, small in (("cunet/art", 2, False), ("swin_unet/art", 4, True)):
        path = sm.model_path(str(root / model.replace("/", "_")), model, scale, 1)
        sm.export_onnx(sm.make_model(model, scale, seed=3, small=small), path, 1, 64)
        out[model] = path
    return out


def test_valid_models_lower_under_the_sanitizers(small_models):
    """both graph families, fp16 and fp32 plans, through load_onnx -> fold_graph -> lower_graph -> serialize -> deserialize"""
    for model, path in small_models.items():
        for extra in ((), ("fp32",)):
            rc, out = check("onnx", path, 1, 64, *extra)
            assert rc == 0 and out.startswith("ok:"), (model, extra, out[-2000:])
        rc, out = check("onnx", path, 2, 64)                      # a static batch of 1 asked for 2: a clean refusal
        assert rc == 2 and "fixed to 1" in out, out


def test_truncated_and_bit_flipped_onnx_files(small_models, tmp_path):
    rng = np.random.default_rng(2024)
    blobs = {}
    for model, path in small_models.items():
        data = open(path, "rb").read()
        tag = model.split("/")[0] + "_"
        # the node list and the shape constants sit in front of the weight initialisers: most flips go there
        blobs.update(mutants(data, rng, 12, 28, region=(0, len(data) // 5), tag=tag + "head_"))
        blobs.update(mutants(data, rng, 0, 8, tag=tag + "any_"))
    blobs["empty"] = b""
    blobs["not_protobuf"] = b"\x89PNG\r\n\x1a\n" + bytes(range(256)) * 8
    blobs["varint_overrun"] = b"\x3a" + b"\xff" * 64                                  # graph field whose length varint never ends
    blobs["huge_length"] = b"\x3a\xff\xff\xff\xff\x0f" + b"\x00" * 32                 # graph field claiming 4 GB
    blobs["nested_depth"] = b"\x3a\x10" * 2000                                        # length-delimited nesting that runs out of bytes
    res = run_corpus("onnx", write_all(tmp_path, blobs, ".onnx"), extra=(1, 64))
    assert res["empty"] == 2 and res["not_protobuf"] == 2 and res["huge_length"] == 2
    assert sum(rc == 2 for rc in res.values()) >= 20, "the corpus does not reach the rejection paths"


# ---- well-formed protobuf, hostile content: a minimal ONNX writer (wire format of SURVEY.md appendix C; no onnx package offline)
def _vint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F; n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(field, payload): return _vint(field << 3 | 2) + _vint(len(payload)) + payload
def _iv(field, v): return _vint(field << 3 | 0) + _vint(v)
def _tensor(name, dtype, dims, raw): return b"".join(_iv(1, d) for d in dims) + _iv(2, dtype) + _ld(8, name.encode()) + _ld(9, raw)
def _attr_t(name, tensor): return _ld(1, name.encode()) + _ld(5, tensor) + _iv(20, 4)
def _attr_i(name, v): return _ld(1, name.encode()) + _iv(3, v) + _iv(20, 2)
def _attr_ints(name, vals): return _ld(1, name.encode()) + b"".join(_iv(8, v) for v in vals) + _iv(20, 7)
def _node(op, ins, outs, attrs=(), name=""): return b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs) + _ld(3, (name or op).encode()) + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)
def _vinfo(name, dims): return _ld(1, name.encode()) + _ld(2, _ld(1, _iv(1, 1) + _ld(2, b"".join(_ld(1, _iv(1, d)) for d in dims))))
def _i64s(vals): return np.array(vals, "<i8").tobytes()


def _model(nodes, inits=(), in_dims=(1, 3, 64, 64), out_dims=(1, 3, 64, 64), out="y"):
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in inits) + _ld(11, _vinfo("x", in_dims)) + _ld(12, _vinfo(out, out_dims))
    return _iv(1, 8) + _ld(2, b"hostile") + _ld(7, g) + _ld(8, _iv(2, 17))


def test_well_formed_but_hostile_onnx_graphs(tmp_path):
    """Files the protobuf reader accepts whose CONTENT asks for the impossible: constants of 2^40 elements (ConstantOfShape, Expand, Range: the folding pass
    evaluates whatever does not depend on the input), initialisers whose dims do not match their payload or are negative, a node that reads its own output, an
    undefined input, a rank-5 graph input, an operator nobody lowers.  Each must be refused with a message; none may allocate what the file dictates."""
    huge = 1 << 40
    one = _tensor("", 1, [1], np.float32([1.0]).tobytes())
    blobs = {
        "const_of_shape_huge": _model([_node("ConstantOfShape", ["s"], ["c"], [_attr_t("value", one)]), _node("Add", ["x", "c"], ["y"])], [_tensor("s", 7, [1], _i64s([huge]))]),
        "const_of_shape_negative": _model([_node("ConstantOfShape", ["s"], ["c"]), _node("Add", ["x", "c"], ["y"])], [_tensor("s", 7, [2], _i64s([-5, 7]))]),
        "expand_huge": _model([_node("Expand", ["one", "s"], ["c"]), _node("Add", ["x", "c"], ["y"])], [_tensor("one", 1, [1], np.float32([1]).tobytes()), _tensor("s", 7, [2], _i64s([1 << 20, 1 << 20]))]),
        "range_huge": _model([_node("Range", ["a", "b", "d"], ["c"]), _node("Add", ["x", "c"], ["y"])], [_tensor("a", 7, [], _i64s([0])), _tensor("b", 7, [], _i64s([huge])), _tensor("d", 7, [], _i64s([1]))]),
        "range_zero_step": _model([_node("Range", ["a", "b", "d"], ["c"]), _node("Add", ["x", "c"], ["y"])], [_tensor("a", 7, [], _i64s([0])), _tensor("b", 7, [], _i64s([9])), _tensor("d", 7, [], _i64s([0]))]),
        "init_dims_exceed_payload": _model([_node("Add", ["x", "w"], ["y"])], [_tensor("w", 1, [1 << 30, 1 << 30], np.float32([1, 2]).tobytes())]),
        "init_dims_exceed_payload_f16": _model([_node("Add", ["x", "w"], ["y"])], [_tensor("w", 10, [1 << 31, 4], np.float16([1, 2]).tobytes())]),
        "init_dims_exceed_payload_i32": _model([_node("Add", ["x", "w"], ["y"])], [_tensor("w", 6, [1 << 40], np.int32([1, 2]).tobytes())]),
        "init_negative_dim": _model([_node("Add", ["x", "w"], ["y"])], [_tensor("w", 1, [-1, 4], np.float32([1, 2, 3, 4]).tobytes())]),
        "init_dims_overflow": _model([_node("Add", ["x", "w"], ["y"])], [_tensor("w", 1, [1 << 62, 1 << 62, 4], b"")]),
        "node_reads_itself": _model([_node("Relu", ["y"], ["y"])]),
        "undefined_input": _model([_node("Add", ["x", "nowhere"], ["y"])]),
        "rank5_input": _model([_node("Relu", ["x"], ["y"])], in_dims=(1, 3, 4, 64, 64), out_dims=(1, 3, 4, 64, 64)),
        "zero_sized_input": _model([_node("Relu", ["x"], ["y"])], in_dims=(1, 0, 64, 64), out_dims=(1, 0, 64, 64)),
        "unknown_operator": _model([_node("NonMaxSuppression", ["x"], ["y"])]),
        "reshape_to_huge": _model([_node("Reshape", ["x", "s"], ["y"])], [_tensor("s", 7, [4], _i64s([huge, huge, huge, -1]))]),
        "conv_kernel_huge": _model([_node("Conv", ["x", "w"], ["y"], [_attr_i("group", 1)])], [_tensor("w", 1, [1 << 20, 3, 1 << 20, 3], b"")]),
        "no_nodes": _model([], out="x"),
        # round 5: what the graph simplifier (csrc/simplify.cpp) rewrites - each in a form it must leave alone or refuse, never index out of range
        "transpose_perm_out_of_range": _model([_node("Transpose", ["x"], ["y"], [_attr_ints("perm", [0, 7, 2, 3])])]),
        "transpose_perm_repeats": _model([_node("Transpose", ["x"], ["t"], [_attr_ints("perm", [0, 1, 1, 3])]), _node("Transpose", ["t"], ["y"], [_attr_ints("perm", [0, 1, 2, 3])])]),
        "transpose_perm_short": _model([_node("Transpose", ["x"], ["t"], [_attr_ints("perm", [1, 0])]), _node("Transpose", ["t"], ["y"], [_attr_ints("perm", [0, 3, 1, 2])])]),
        "identity_input_to_output": _model([_node("Identity", ["x"], ["y"])]),
        "gemm_transA": _model([_node("Reshape", ["x", "s"], ["x2"]), _node("Gemm", ["x2", "w"], ["y2"], [_attr_i("transA", 1)]), _node("Reshape", ["y2", "s4"], ["y"])],
                              [_tensor("s", 7, [2], _i64s([-1, 64])), _tensor("w", 1, [192, 64], np.zeros(192 * 64, np.float32).tobytes()), _tensor("s4", 7, [4], _i64s([1, 3, 64, 64]))]),
        "gemm_rank3_weight": _model([_node("Reshape", ["x", "s"], ["x2"]), _node("Gemm", ["x2", "w"], ["y2"]), _node("Reshape", ["y2", "s4"], ["y"])],
                                    [_tensor("s", 7, [2], _i64s([-1, 64])), _tensor("w", 1, [2, 64, 32], np.zeros(4096, np.float32).tobytes()), _tensor("s4", 7, [4], _i64s([1, 3, 64, 64]))]),
        "gemm_bias_wrong_length": _model([_node("Reshape", ["x", "s"], ["x2"]), _node("Gemm", ["x2", "w", "b"], ["y2"]), _node("Reshape", ["y2", "s4"], ["y"])],
                                         [_tensor("s", 7, [2], _i64s([-1, 64])), _tensor("w", 1, [64, 64], np.zeros(4096, np.float32).tobytes()), _tensor("b", 1, [7], np.zeros(7, np.float32).tobytes()),
                                          _tensor("s4", 7, [4], _i64s([1, 3, 64, 64]))]),
        "reshape_chain_to_nothing": _model([_node("Reshape", ["x", "s"], ["a"]), _node("Reshape", ["a", "s4"], ["b"]), _node("Flatten", ["b"], ["c"]), _node("Reshape", ["c", "s4"], ["y"])],
                                           [_tensor("s", 7, [2], _i64s([-1, 64])), _tensor("s4", 7, [4], _i64s([1, 3, 64, 64]))]),
        "cast_to_int_and_back": _model([_node("Cast", ["x"], ["i"], [_attr_i("to", 7)]), _node("Cast", ["i"], ["y"], [_attr_i("to", 1)])]),
        "squeeze_axis_out_of_range": _model([_node("Squeeze", ["x", "ax"], ["y"])], [_tensor("ax", 7, [1], _i64s([9]))]),
        "dead_branch_only": _model([_node("Relu", ["x"], ["unused"]), _node("Identity", ["x"], ["y"])]),
    }
    res = run_corpus("onnx", write_all(tmp_path, blobs, ".onnx"), extra=(1, 64))
    assert all(rc == 2 for k, rc in res.items() if k != "no_nodes"), {k: v for k, v in res.items() if v != 2}      # (an empty graph is the identity network: a plan of zero ops)


def test_truncated_and_bit_flipped_engine_files(pkg, small_models, tmp_path):
    rng = np.random.default_rng(7)
    blobs = {}
    for model, path in small_models.items():
        eng = tmp_path / (model.replace("/", "_") + ".w2x")
        assert pkg.write_engine_file(path, 1, 64, str(eng))
        data = eng.read_bytes()
        tag = model.split("/")[0] + "_"
        blobs[tag + "valid"] = data
        # header + tensor / op tables are a few KB in front of and behind the weight blobs
        blobs.update(mutants(data, rng, 14, 60, region=(0, 4096), tag=tag + "head_"))
        blobs.update(mutants(data, rng, 0, 60, region=(max(len(data) - 16384, 0), len(data)), tag=tag + "tail_"))
        blobs.update(mutants(data, rng, 0, 20, tag=tag + "any_"))
    blobs["empty"] = b""
    blobs["magic_only"] = next(iter(blobs.values()))[:8]
    res = run_corpus("plan", write_all(tmp_path, blobs, ".w2x"))
    assert res["cunet_valid"] == 0 and res["swin_unet_valid"] == 0 and res["empty"] == 2 and res["magic_only"] == 2
    assert sum(rc == 2 for rc in res.values()) >= 30


# ----------------------------------------------------------------------------------------------------------------------------- images
def _chunk(t, body, crc=None):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) if crc is None else crc)


def _png(w, h, depth, ctype, raw, interlace=0, extra=b"", ihdr_dims=None):
    dw, dh = ihdr_dims or (w, h)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", dw, dh, depth, ctype, 0, 0, interlace)) + extra +
            _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b""))


def _rows(arr):          # [h, w*ch] uint8 -> filter-0 scanlines
    return b"".join(b"\x00" + r.tobytes() for r in arr)


def _adam7(arr8):
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = arr8[y0::dy, x0::dx]
        if sub.size:
            raw += b"".join(b"\x00" + row.tobytes() for row in sub)
    return raw


def test_malformed_png_files(tmp_path):
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (21, 30, 3), dtype=np.uint8)
    rgb = _png(30, 21, 8, 2, _rows(a.reshape(21, -1)))
    rgba16 = _png(9, 7, 16, 6, _rows(rng.integers(0, 256, (7, 9 * 8), dtype=np.uint8)))
    pal = _png(16, 4, 4, 3, _rows(rng.integers(0, 256, (4, 8), dtype=np.uint8)), extra=_chunk(b"PLTE", bytes(range(48))))
    adam = _png(30, 21, 8, 2, _adam7(a), interlace=1)
    blobs = {"valid_rgb": rgb, "valid_rgba16": rgba16, "valid_pal4": pal, "valid_adam7": adam}
    for tag, data in (("rgb_", rgb), ("rgba16_", rgba16), ("pal_", pal), ("adam7_", adam)):
        blobs.update(mutants(data, rng, 12, 40, tag=tag))
    raw = _rows(a.reshape(21, -1))
    blobs.update({
        "huge_ihdr": _png(30, 21, 8, 2, raw, ihdr_dims=(0x7FFFFFFF, 0x7FFFFFFF)),            # 2^62 pixels announced, 2 KB of data
        "wide_ihdr": _png(30, 21, 8, 2, raw, ihdr_dims=(0x10000000, 1)),
        "tall_ihdr": _png(30, 21, 8, 2, raw, ihdr_dims=(1, 0x40000000)),
        "zero_dims": _png(30, 21, 8, 2, raw, ihdr_dims=(0, 0)),
        "bad_depth": _png(30, 21, 3, 2, raw),
        "bad_ctype": _png(30, 21, 8, 5, raw),
        "bad_interlace": _png(30, 21, 8, 2, raw, interlace=2),
        "short_idat": _png(30, 21, 8, 2, raw[:len(raw) // 2]),
        "bad_filter": _png(30, 21, 8, 2, b"\x07" + raw[1:]),
        "bad_zlib": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 30, 21, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", b"\x78\x9c" + bytes(200)) + _chunk(b"IEND", b""),
        "no_idat": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 30, 21, 8, 2, 0, 0, 0)) + _chunk(b"IEND", b""),
        "chunk_len_overflow": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 30, 21, 8, 2, 0, 0, 0)) + b"\xff\xff\xff\xf0IDAT" + bytes(64),
        "palette_missing": _png(16, 4, 4, 3, _rows(rng.integers(0, 256, (4, 8), dtype=np.uint8))),
        "palette_short": _png(16, 4, 8, 3, _rows(np.full((4, 16), 200, np.uint8)), extra=_chunk(b"PLTE", bytes(9))),   # index 200 of a 3-entry palette
        "adam7_short": _png(30, 21, 8, 2, _adam7(a)[:300], interlace=1),
        "ihdr_short": b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", b"\x00\x00\x00\x10"),
        "signature_only": b"\x89PNG\r\n\x1a\n",
        "zip_bomb_rows": _png(30, 21, 8, 2, bytes(50_000_000)),                                    # far more scanline data than the header allows
    })
    files = write_all(tmp_path, blobs, ".png")
    res = run_corpus("image", files)
    for k in ("valid_rgb", "valid_rgba16", "valid_pal4", "valid_adam7"):
        assert res[k] == 0, k
    for k in ("huge_ihdr", "wide_ihdr", "tall_ihdr", "zero_dims", "bad_depth", "bad_ctype", "bad_interlace", "short_idat", "bad_filter", "bad_zlib",
              "no_idat", "chunk_len_overflow", "palette_missing", "adam7_short", "ihdr_short", "signature_only"):
        assert res[k] == 2, k
    # --deep (16-bit samples kept) takes the same files
    res16 = run_corpus("image", {k: v for k, v in files.items() if k.startswith(("valid_rgba16", "rgba16_"))}, extra=("deep",))
    assert res16["valid_rgba16"] == 0


def _bmp(w, h, bpp, body, off=54, size_field=None, comp=0, planes=1):
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(body) if size_field is None else size_field, 0, 0, off)
    dib = struct.pack("<IiiHHIIiiII", 40, w, h, planes, bpp, comp, len(body), 2835, 2835, 0, 0)
    return hdr + dib + body


def test_malformed_bmp_and_ppm_files(tmp_path):
    rng = np.random.default_rng(13)
    body24 = rng.integers(0, 256, 12 * ((13 * 3 + 3) // 4 * 4), dtype=np.uint8).tobytes()
    body32 = rng.integers(0, 256, 12 * 13 * 4, dtype=np.uint8).tobytes()
    v24, v32, vtd = _bmp(13, 12, 24, body24), _bmp(13, 12, 32, body32), _bmp(13, -12, 24, body24)
    blobs = {"valid24": v24, "valid32": v32, "valid_topdown": vtd}
    for tag, data in (("b24_", v24), ("b32_", v32)):
        blobs.update(mutants(data, rng, 12, 60, region=(0, 54), tag=tag))
    blobs.update({
        "huge": _bmp(0x7FFFFFFF, 0x7FFFFFFF, 24, body24), "wide": _bmp(0x20000000, 1, 24, body24), "neg_width": _bmp(-13, 12, 24, body24),
        "min_height": _bmp(13, -0x80000000, 24, body24), "zero": _bmp(0, 0, 24, b""), "bpp0": _bmp(13, 12, 0, body24), "bpp16": _bmp(13, 12, 16, body24),
        "rle": _bmp(13, 12, 24, body24, comp=1), "off_past_end": _bmp(13, 12, 24, body24, off=1 << 30), "off_in_header": _bmp(13, 12, 24, body24, off=10),
        "short_body": _bmp(13, 12, 24, body24[:100]), "header_only": v24[:54], "bm": b"BM",
    })
    res = run_corpus("image", write_all(tmp_path, blobs, ".bmp"))
    assert res["valid24"] == 0 and res["valid32"] == 0 and res["valid_topdown"] == 0
    for k in ("huge", "wide", "neg_width", "min_height", "zero", "bpp0", "bpp16", "rle", "off_past_end", "short_body", "header_only", "bm"):
        assert res[k] == 2, k
    ppm = b"P6\n13 12\n255\n" + bytes(13 * 12 * 3)
    pblobs = {"valid": ppm, "short": ppm[:100], "huge": b"P6\n2000000000 2000000000\n255\n" + bytes(64), "neg": b"P6\n-3 4\n255\n" + bytes(64), "maxval": b"P6\n13 12\n65535\n" + bytes(13 * 12 * 6),
              "p3": b"P3\n1 1\n255\n1 2 3\n", "no_dims": b"P6\n", "comment_forever": b"P6\n#" + b"x" * 5000, "digits_forever": b"P6\n" + b"9" * 5000}
    pblobs.update(mutants(ppm, np.random.default_rng(5), 8, 30, region=(0, 16), tag="ppm_"))
    pres = run_corpus("image", write_all(tmp_path, pblobs, ".ppm"))
    assert pres["valid"] == 0 and pres["short"] == 2 and pres["huge"] == 2 and pres["neg"] == 2 and pres["no_dims"] == 2


def _avi(frames, w, h, fps=25):
    """the uncompressed RIFF AVI the built-in writer emits (tests/test_cli.py _write_avi), as bytes"""
    stride = (w * 3 + 3) // 4 * 4
    def ck(t, b): return t + struct.pack("<I", len(b)) + b + (b"\x00" if len(b) & 1 else b"")
    def lst(t, b): return b"LIST" + struct.pack("<I", len(b) + 4) + t + b
    img = stride * h
    avih = struct.pack("<14I", 1000000 // fps, img * fps, 0, 0x10, len(frames), 0, 1, img, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"DIB " + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, len(frames), img, 0xFFFFFFFF, 0) + struct.pack("<4h", 0, 0, w, h)
    strf = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, img, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", ck(b"avih", avih) + lst(b"strl", ck(b"strh", strh) + ck(b"strf", strf)))
    movi = b""
    for f in frames:
        rows = b"".join(f[y].tobytes() + bytes(stride - w * 3) for y in range(h - 1, -1, -1))
        movi += ck(b"00db", rows)
    body = b"AVI " + hdrl + lst(b"movi", movi)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_malformed_avi_files(tmp_path):
    rng = np.random.default_rng(17)
    frames = [rng.integers(0, 256, (10, 14, 3), dtype=np.uint8) for _ in range(3)]
    valid = _avi(frames, 14, 10)
    hdr_end = valid.index(b"movi")
    blobs = {"valid": valid}
    blobs.update(mutants(valid, rng, 16, 120, region=(0, hdr_end + 16), tag="hdr_"))
    blobs.update(mutants(valid, rng, 0, 30, tag="any_"))
    big = bytearray(valid)
    p = valid.index(b"strf") + 8
    big[p + 4:p + 12] = struct.pack("<ii", 0x7FFFFFFF, 0x7FFFFFFF)
    blobs["huge_dims"] = bytes(big)
    neg = bytearray(valid); neg[p + 4:p + 8] = struct.pack("<i", -14); blobs["neg_width"] = bytes(neg)
    z = bytearray(valid); z[p + 4:p + 12] = struct.pack("<ii", 0, 0); blobs["zero_dims"] = bytes(z)
    c = bytearray(valid); q = valid.index(b"00db") + 4; c[q:q + 4] = struct.pack("<I", 0xFFFFFFF0); blobs["chunk_past_end"] = bytes(c)
    blobs["riff_only"] = valid[:12]
    blobs["empty"] = b""
    res = run_corpus("avi", write_all(tmp_path, blobs, ".avi"))
    assert res["valid"] == 0
    for k in ("huge_dims", "neg_width", "zero_dims", "riff_only", "empty"):
        assert res[k] == 2, k


def test_option_parser_and_tile_grid_under_the_sanitizers(tmp_path):
    base = ["--model", "swin_unet/art", "--scale", "4", "--noise", "3", "--batchSize", "4", "--tileSize", "256"]
    img = tmp_path / "a.png"; img.write_bytes(b"x")
    cases = [base + ["render", "-i", str(img)], base + ["build"], ["render"], [], base + ["render", "-i"], base + ["--scale"], ["--scale=", "build"],
             base + ["render", "-i", str(img), "--blend", "1/0"], base + ["render", "-i", str(img), "--blend", "/"], base + ["render", "-i", str(img), "--crf", "99999999999999999999"],
             ["--batchSize", "-2147483648"] + base + ["build"], base + ["--device", "9" * 40, "build"], base + ["render", "-i", str(img), "--tta-mode", ""],
             ["--model=" + "x" * 100000, "build"], base + ["render", "-i", str(img)] + ["-i", str(img)] * 500, ["--", "--", "--"], ["-i"], ["=", "build"]]
    for args in cases:
        rc, out = check("args", *args)
        assert rc in (0, 2), (args[:6], out[-1500:])
    assert check("args", *base, "render", "-i", str(img))[0] == 0
    # the tile grid and the strip plans for every BASELINE configuration and a few degenerate frames (img2img_render.cpp:7-66)
    for w, h, t, s, to, ov in [(1920, 1080, 256, 4, 960, 0.0625), (3840, 2160, 640, 4, 2496, 0.0625), (256, 256, 64, 2, 56, 0.0625), (1, 1, 64, 2, 56, 0.125),
                               (7, 3000, 64, 4, 192, 0.03125), (1920, 1080, 400, 4, 1536, 0.0), (65, 65, 64, 1, 28, 0.125)]:
        rc, out = check("tiles", w, h, t, s, to, ov, ov)
        assert rc == 0, (w, h, t, out[-1500:])
