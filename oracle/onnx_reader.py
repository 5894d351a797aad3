"""ORACLE (test infrastructure, not product code): minimal ONNX protobuf reader.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product has its own independent C++ reader (waifu2x-tensorrt_amd/csrc/onnx_pb.cpp).

The reference hands the ONNX file to TensorRT's parser (src/tensorrt/img2img_build.cpp:81-88,
nvonnxparser::IParser::parseFromFile); neither that parser nor the `onnx` python package is
available offline, so the wire format is decoded by hand (field numbers: SURVEY.md Appendix C).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np


def _varint(buf, pos):
    r = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) for one message body (bytes/memoryview)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fn, wt, v


def _sint64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_or_single_int(wt, v, out):
    if wt == 0:
        out.append(_sint64(v))
    else:  # packed
        pos = 0
        while pos < len(v):
            x, pos = _varint(v, pos)
            out.append(_sint64(x))


_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_,
           10: np.float16, 11: np.float64}


def _tensor(buf):
    dims, dtype, raw, name = [], 1, None, ""
    f32, i32, i64, f64 = [], [], [], []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            _packed_or_single_int(wt, v, dims)
        elif fn == 2:
            dtype = v
        elif fn == 4:
            if wt == 5:
                f32.append(struct.unpack("<f", v)[0])
            else:
                f32.extend(np.frombuffer(bytes(v), "<f4").tolist())
        elif fn == 5:
            _packed_or_single_int(wt, v, i32)
        elif fn == 7:
            _packed_or_single_int(wt, v, i64)
        elif fn == 10:
            if wt == 1:
                f64.append(struct.unpack("<d", v)[0])
            else:
                f64.extend(np.frombuffer(bytes(v), "<f8").tolist())
        elif fn == 8:
            name = bytes(v).decode()
        elif fn == 9:
            raw = bytes(v)
        elif fn == 14 and v == 1:
            raise ValueError("external tensor data is not supported")
    np_dt = _DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dt).newbyteorder("<")).astype(np_dt)
    elif dtype == 1:
        arr = np.asarray(f32, np.float32)
    elif dtype == 7:
        arr = np.asarray(i64, np.int64)
    elif dtype == 11:
        arr = np.asarray(f64, np.float64)
    elif dtype == 10:
        arr = np.asarray(i32, np.uint16).view(np.float16)
    else:
        arr = np.asarray(i32).astype(np_dt)
    return name, arr.reshape(dims).copy()


def _attr(buf):
    name, val, typ = "", None, 0
    f = i = s = t = None
    floats, ints, strings = [], [], []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            name = bytes(v).decode()
        elif fn == 2:
            f = struct.unpack("<f", v)[0]
        elif fn == 3:
            i = _sint64(v)
        elif fn == 4:
            s = bytes(v)
        elif fn == 5:
            t = _tensor(v)[1]
        elif fn == 7:
            if wt == 5:
                floats.append(struct.unpack("<f", v)[0])
            else:
                floats.extend(np.frombuffer(bytes(v), "<f4").tolist())
        elif fn == 8:
            _packed_or_single_int(wt, v, ints)
        elif fn == 9:
            strings.append(bytes(v))
        elif fn == 20:
            typ = v
    if typ == 1:
        val = f
    elif typ == 2:
        val = i
    elif typ == 3:
        val = s.decode()
    elif typ == 4:
        val = t
    elif typ == 6:
        val = floats
    elif typ == 7:
        val = ints
    elif typ == 8:
        val = [x.decode() for x in strings]
    else:  # untyped writer: pick whatever is present
        val = t if t is not None else (ints or floats or i if i is not None else f)
    return name, val


@dataclass
class Node:
    op: str
    inputs: list
    outputs: list
    attrs: dict
    name: str = ""


@dataclass
class ValueInfo:
    name: str
    elem_type: int
    shape: list  # int or str (dim_param)


@dataclass
class Graph:
    nodes: list = field(default_factory=list)
    initializers: dict = field(default_factory=dict)
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)
    opset: int = 0


def _value_info(buf):
    name, et, shape = "", 0, []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            name = bytes(v).decode()
        elif fn == 2:
            for fn2, _, v2 in _fields(v):
                if fn2 == 1:  # tensor_type
                    for fn3, _, v3 in _fields(v2):
                        if fn3 == 1:
                            et = v3
                        elif fn3 == 2:
                            for fn4, _, v4 in _fields(v3):
                                if fn4 == 1:
                                    d = None
                                    for fn5, _, v5 in _fields(v4):
                                        if fn5 == 1:
                                            d = _sint64(v5)
                                        elif fn5 == 2:
                                            d = bytes(v5).decode()
                                    shape.append(d)
    return ValueInfo(name, et, shape)


def _node(buf):
    ins, outs, attrs, op, name = [], [], {}, "", ""
    for fn, wt, v in _fields(buf):
        if fn == 1:
            ins.append(bytes(v).decode())
        elif fn == 2:
            outs.append(bytes(v).decode())
        elif fn == 3:
            name = bytes(v).decode()
        elif fn == 4:
            op = bytes(v).decode()
        elif fn == 5:
            k, val = _attr(v)
            attrs[k] = val
    return Node(op, ins, outs, attrs, name)


def load(path: str) -> Graph:
    with open(path, "rb") as fh:
        data = memoryview(fh.read())
    g = Graph()
    for fn, wt, v in _fields(data):
        if fn == 7:  # graph
            for fn2, _, v2 in _fields(v):
                if fn2 == 1:
                    g.nodes.append(_node(v2))
                elif fn2 == 5:
                    n, a = _tensor(v2)
                    g.initializers[n] = a
                elif fn2 == 11:
                    g.inputs.append(_value_info(v2))
                elif fn2 == 12:
                    g.outputs.append(_value_info(v2))
        elif fn == 8:  # opset_import
            dom, ver = "", 0
            for fn2, _, v2 in _fields(v):
                if fn2 == 1:
                    dom = bytes(v2).decode()
                elif fn2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                g.opset = ver
    g.inputs = [vi for vi in g.inputs if vi.name not in g.initializers]
    return g
