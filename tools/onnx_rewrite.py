"""Test infrastructure: an ONNX writer and a set of semantics-preserving graph rewrites.

The reference hands whatever ONNX file it is given to TensorRT's parser (src/tensorrt/img2img_build.cpp:81-88); the files its
users have are nunif exports this repository cannot see (README.md:11-15, no network).  What CAN be done offline: take the graphs
tools/synth_models.py exports and re-spell them the ways exporters, opsets and optimiser passes spell the same computation - and
require the loader (csrc/fold.cpp, simplify.cpp, lower.cpp) to produce the plan of the original for every spelling
(tests/test_loader_rewrites.py).

    g = oracle.onnx_reader.load(path); shapes = runtime_shapes(path)
    v = rewrite(g, shapes, seed)          # a deep copy with 2-5 rewrites applied at random sites; v.applied lists them
    dump(v, out_path)

Rewrites (each keeps the function the graph computes, bit for bit in fp32 arithmetic, except `fp16_init`, which stores weights
rounded to fp16 - the precision an fp16 engine gives them anyway):
    gemm            MatMul(x, W) + Add(b)  ->  Reshape([-1, K]) -> Gemm(transB at random) -> Reshape(lead.., N)   (shape via Shape/Slice/Concat or a constant)
    identity        Identity on an edge
    dropout         Dropout (inference: identity) on an edge
    cast            Cast(to=FLOAT) on a float edge
    transpose2      Transpose(p) -> Transpose(p^-1) on an edge
    squeeze         Unsqueeze(axis) -> Squeeze(axis) on an edge  (Flatten -> Reshape back for 4-D edges)
    const_node      an initializer becomes a Constant node
    fp16_init       a float initializer is stored as FLOAT16 and widened by a Cast node
    reshape_0_m1    a Reshape target is re-spelled with 0 (copy the input dimension) and / or -1 (infer)
    bias_unsqueeze  a bias vector [N] is stored as [N] and reaches its Add through Unsqueeze to [1, 1, N] (or stored [1, N] and squeezed)
    ln_axis         LayerNormalization axis -1 <-> rank - 1
    permute         the node list in another topological order
    commute         the operands of an Add / Mul swapped
    split_qkv       q, k, v = qkv[0], qkv[1], qkv[2] (three Gathers)  ->  qkv.unbind(0) (one Split + three Squeezes)
    ln_decompose    a LayerNormalization node  ->  the ReduceMean / Sub / Pow|Mul / ReduceMean / Add / Sqrt / Div / Mul / Add chain exporters write below opset 17
    sdpa_scale      q * s  ->  (q * sqrt(s)) @ (k^T * sqrt(s)), the way a decomposed scaled_dot_product_attention scales (not bit-exact: sqrt(s)^2 != s in fp32)
    d2s_dcr         DepthToSpace(mode=CRD) behind a Linear  ->  mode=DCR (or no mode attribute: DCR is the operator's default) with the Linear's output columns and bias
                    re-ordered so that the same values land on the same sub-pixels (what an exporter that writes DCR produces for the same network)
    gelu_op         the erf chain Div(sqrt 2) -> Erf -> Add(1) -> Mul(x) -> Mul(0.5)  ->  one Gelu node, as the exporter writes it from opset 20 on (the file's opset becomes 20)
    dead            a node nothing reads (a Shape or a Relu of some runtime tensor)
"""
from __future__ import annotations

import copy
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import onnx_reader  # noqa: E402
from oracle.onnx_reader import Graph, Node  # noqa: E402

# ------------------------------------------------------------------------------------------------ protobuf writer


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(fn: int, wt: int) -> bytes:
    return _varint((fn << 3) | wt)


def _ld(fn: int, payload: bytes) -> bytes:
    return _key(fn, 2) + _varint(len(payload)) + payload


def _vi(fn: int, v: int) -> bytes:
    return _key(fn, 0) + _varint(v)


_NP2ONNX = {np.dtype(np.float32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3, np.dtype(np.int32): 6, np.dtype(np.int64): 7,
            np.dtype(np.bool_): 9, np.dtype(np.float16): 10, np.dtype(np.float64): 11}


def _tensor(name: str, a: np.ndarray, packed_dims: bool) -> bytes:
    a = np.asarray(a)
    out = b""
    if packed_dims and a.ndim:
        out += _ld(1, b"".join(_varint(d) for d in a.shape))          # the official writer packs repeated scalars ...
    else:
        out += b"".join(_vi(1, d) for d in a.shape)                   # ... torch's legacy exporter writes one tagged varint each
    out += _vi(2, _NP2ONNX[a.dtype])
    if name:
        out += _ld(8, name.encode())
    out += _ld(9, np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes())
    return out


def _attr(name: str, v, packed: bool) -> bytes:
    out = _ld(1, name.encode())
    if isinstance(v, np.ndarray):
        out += _ld(5, _tensor("", v, packed)) + _vi(20, 4)
    elif isinstance(v, bool):
        out += _vi(3, int(v)) + _vi(20, 2)
    elif isinstance(v, (int, np.integer)):
        out += _vi(3, int(v)) + _vi(20, 2)
    elif isinstance(v, float):
        out += _key(2, 5) + struct.pack("<f", v) + _vi(20, 1)
    elif isinstance(v, str):
        out += _ld(4, v.encode()) + _vi(20, 3)
    elif isinstance(v, (list, tuple)) and all(isinstance(e, float) for e in v) and v:
        out += (_ld(7, b"".join(struct.pack("<f", e) for e in v)) if packed else b"".join(_key(7, 5) + struct.pack("<f", e) for e in v)) + _vi(20, 6)
    elif isinstance(v, (list, tuple)) and all(isinstance(e, str) for e in v) and v:
        out += b"".join(_ld(9, e.encode()) for e in v) + _vi(20, 8)
    elif isinstance(v, (list, tuple)):
        out += (_ld(8, b"".join(_varint(int(e)) for e in v)) if packed and v else b"".join(_vi(8, int(e)) for e in v)) + _vi(20, 7)
    else:
        raise TypeError(f"attribute {name}: {type(v)}")
    return out


def _value_info(vi) -> bytes:
    dims = b""
    for d in vi.shape:
        dims += _ld(1, _ld(2, d.encode()) if isinstance(d, str) else _vi(1, int(d)))
    tensor_type = _vi(1, vi.elem_type) + _ld(2, dims)
    return _ld(1, vi.name.encode()) + _ld(2, _ld(1, tensor_type))


def dump(g: Graph, path: str, packed: bool = False, producer: str = "w2x-rewrite", ir_version: int = 8) -> None:
    """Write `g` as an ONNX ModelProto.  packed: repeated integers / floats as one length-delimited blob (what the `onnx` package writes) instead of one
    tagged scalar per element (what torch's legacy exporter writes) - the reader must take both."""
    parts = []
    for n in g.nodes:
        nb = b"".join(_ld(1, i.encode()) for i in n.inputs) + b"".join(_ld(2, o.encode()) for o in n.outputs)
        if n.name:
            nb += _ld(3, n.name.encode())
        nb += _ld(4, n.op.encode())
        for k, v in n.attrs.items():
            nb += _ld(5, _attr(k, v, packed))
        parts.append(_ld(1, nb))
    parts.append(_ld(2, b"main_graph"))
    for name, a in g.initializers.items():
        parts.append(_ld(5, _tensor(name, a, packed)))
    for vi in g.inputs:
        parts.append(_ld(11, _value_info(vi)))
    for vi in g.outputs:
        parts.append(_ld(12, _value_info(vi)))
    graph = b"".join(parts)
    model = _vi(1, ir_version) + _ld(2, producer.encode()) + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, g.opset))
    with open(path, "wb") as f:
        f.write(model)


# ------------------------------------------------------------------------------------------------ shapes of the runtime tensors


def runtime_shapes(path: str, batch: int, tile: int) -> dict:
    """name -> shape of every float tensor that depends on the graph input's DATA (Shape / Size outputs and what is computed from them are constants at a
    static input shape), from one run of the oracle's executor on a zero tile."""
    from oracle import onnx_exec
    ex = onnx_exec.Executor(path)
    g = ex.g
    runtime = {g.inputs[0].name}
    for n in g.nodes:
        if n.op not in ("Shape", "Size") and any(i in runtime for i in n.inputs if i):
            runtime.update(o for o in n.outputs if o)
    static = [o for n in g.nodes for o in n.outputs if o and o not in runtime]
    vals = ex.run(np.zeros((batch, 3, tile, tile), np.float32), keep=tuple(runtime) + tuple(static))
    shapes = {k: tuple(int(d) for d in v.shape) for k, v in vals.items() if k in runtime and v.dtype.kind == "f"}      # float tensors only (not the Shape -> ... integer chains)
    shapes[g.inputs[0].name] = (batch, 3, tile, tile)
    # scalars the graph computes from constants (an attention scale traced as Pow(head_dim, -0.5)): name -> value, under a key no tensor has
    shapes["__static_scalars__"] = {k: np.asarray(vals[k]) for k in static if k in vals and np.asarray(vals[k]).size == 1}
    return shapes


# ------------------------------------------------------------------------------------------------ rewrites


class Variant(Graph):
    applied: list


def _fresh(g, stem: str) -> str:
    g._n = getattr(g, "_n", 0) + 1
    return f"/rw/{stem}_{g._n}"


def _runtime_edges(g, shapes):
    """(consumer node index, input slot, tensor name) of every runtime float tensor read by a node."""
    out = []
    for k, n in enumerate(g.nodes):
        for s, i in enumerate(n.inputs):
            if i in shapes and not i.startswith("__"):
                out.append((k, s, i))
    return out


def _insert_on_edge(g, shapes, rng, make_nodes, min_rank=1, want_rank=None):
    """Put the nodes make_nodes(src, dst, shape) returns between a random producer and ONE of its consumers."""
    edges = [e for e in _runtime_edges(g, shapes) if len(shapes[e[2]]) >= min_rank and (want_rank is None or len(shapes[e[2]]) == want_rank)]
    if not edges:
        return False
    k, s, name = edges[rng.integers(len(edges))]
    dst = _fresh(g, "e")
    new = make_nodes(name, dst, shapes[name])
    for n in new:
        for o in n.outputs:
            shapes.setdefault(o, shapes[name])        # (intermediate shapes are only needed for rank decisions of later rewrites: approximate is fine)
    shapes[dst] = shapes[name]
    g.nodes[k].inputs[s] = dst
    g.nodes[k:k] = new
    return True


def rw_identity(g, shapes, rng):
    return _insert_on_edge(g, shapes, rng, lambda a, b, sh: [Node("Identity", [a], [b], {}, _fresh(g, "Identity"))])


def rw_dropout(g, shapes, rng):
    return _insert_on_edge(g, shapes, rng, lambda a, b, sh: [Node("Dropout", [a], [b], {}, _fresh(g, "Dropout"))])


def rw_cast(g, shapes, rng):
    return _insert_on_edge(g, shapes, rng, lambda a, b, sh: [Node("Cast", [a], [b], {"to": 1}, _fresh(g, "Cast"))])


def rw_transpose2(g, shapes, rng):
    def mk(a, b, sh):
        r = len(sh)
        p = [int(v) for v in rng.permutation(r)]
        inv = [p.index(k) for k in range(r)]
        mid = _fresh(g, "t")
        shapes[mid] = tuple(sh[k] for k in p)
        return [Node("Transpose", [a], [mid], {"perm": p}, _fresh(g, "Transpose")), Node("Transpose", [mid], [b], {"perm": inv}, _fresh(g, "Transpose"))]
    return _insert_on_edge(g, shapes, rng, mk, min_rank=2)


def rw_squeeze(g, shapes, rng):
    def mk(a, b, sh):
        mid = _fresh(g, "u")
        if len(sh) == 4 and rng.integers(2):
            shp = _fresh(g, "shape")
            g.initializers[shp] = np.asarray(sh, np.int64)
            shapes[mid] = (sh[0], int(np.prod(sh[1:])))
            return [Node("Flatten", [a], [mid], {"axis": 1}, _fresh(g, "Flatten")), Node("Reshape", [mid, shp], [b], {}, _fresh(g, "Reshape"))]
        ax = int(rng.integers(len(sh) + 1))
        shapes[mid] = tuple(sh[:ax]) + (1,) + tuple(sh[ax:])
        if g.opset >= 13:
            axn = _fresh(g, "axes")
            g.initializers[axn] = np.asarray([ax], np.int64)
            return [Node("Unsqueeze", [a, axn], [mid], {}, _fresh(g, "Unsqueeze")), Node("Squeeze", [mid, axn], [b], {}, _fresh(g, "Squeeze"))]
        return [Node("Unsqueeze", [a], [mid], {"axes": [ax]}, _fresh(g, "Unsqueeze")), Node("Squeeze", [mid], [b], {"axes": [ax]}, _fresh(g, "Squeeze"))]
    return _insert_on_edge(g, shapes, rng, mk)


def _consumers(g, name):
    return [(k, n) for k, n in enumerate(g.nodes) if name in n.inputs]


def rw_gemm(g, shapes, rng):
    """MatMul(x, W[K, N]) + Add(b[N]) -> 2-D sandwich around a Gemm."""
    sites = []
    for k, n in enumerate(g.nodes):
        if n.op == "MatMul" and n.inputs[0] in shapes and n.inputs[1] in g.initializers and g.initializers[n.inputs[1]].ndim == 2 and len(shapes[n.inputs[0]]) >= 3:
            us = _consumers(g, n.outputs[0])
            if len(us) == 1 and us[0][1].op == "Add":
                other = [i for i in us[0][1].inputs if i != n.outputs[0]]
                if len(other) == 1 and other[0] in g.initializers and g.initializers[other[0]].ndim == 1:
                    sites.append((k, us[0][0], other[0]))
    if not sites:
        return False
    k, ka, bias = sites[rng.integers(len(sites))]
    mm, add = g.nodes[k], g.nodes[ka]
    x, w = mm.inputs
    K, N = g.initializers[w].shape
    sh = shapes[x]
    two = _fresh(g, "shape2d"); g.initializers[two] = np.asarray([-1, K], np.int64)
    x2, y2 = _fresh(g, "x2d"), _fresh(g, "y2d")
    shapes[x2] = (int(np.prod(sh[:-1])), K); shapes[y2] = (shapes[x2][0], N)
    new = [Node("Reshape", [x, two], [x2], {}, _fresh(g, "Reshape"))]
    attrs = {"alpha": 1.0, "beta": 1.0}
    wname = w
    if rng.integers(2):      # torch.nn.Linear exports its weight as [N, K] with transB = 1
        wname = _fresh(g, "wT"); g.initializers[wname] = np.ascontiguousarray(g.initializers[w].T)
        attrs["transB"] = 1
    new.append(Node("Gemm", [x2, wname, bias], [y2], attrs, mm.name))           # the product keeps the MatMul's name
    if rng.integers(2):      # output shape as a constant ...
        tgt = _fresh(g, "shapeNd"); g.initializers[tgt] = np.asarray(list(sh[:-1]) + [N], np.int64)
    else:                    # ... or computed from the input's shape, as a tracing exporter writes it
        s0, lead, tgt = _fresh(g, "shape"), _fresh(g, "lead"), _fresh(g, "shapeNd")
        st, en, ax, nn = _fresh(g, "c"), _fresh(g, "c"), _fresh(g, "c"), _fresh(g, "c")
        g.initializers[st] = np.asarray([0], np.int64); g.initializers[en] = np.asarray([-1], np.int64)
        g.initializers[ax] = np.asarray([0], np.int64); g.initializers[nn] = np.asarray([N], np.int64)
        new += [Node("Shape", [x], [s0], {}, _fresh(g, "Shape")), Node("Slice", [s0, st, en, ax], [lead], {}, _fresh(g, "Slice")),
                Node("Concat", [lead, nn], [tgt], {"axis": 0}, _fresh(g, "Concat"))]
    new.append(Node("Reshape", [y2, tgt], [add.outputs[0]], {}, _fresh(g, "Reshape")))
    for idx in sorted((k, ka), reverse=True):
        del g.nodes[idx]
    g.nodes[k:k] = new
    if not _consumers(g, w) and w in g.initializers and wname != w:
        del g.initializers[w]
    return True


def rw_const_node(g, shapes, rng):
    names = [n for n in g.initializers if not n.startswith("/rw/")]
    if not names:
        return False
    name = names[rng.integers(len(names))]
    first = min((k for k, n in enumerate(g.nodes) if name in n.inputs), default=None)
    if first is None:
        return False
    a = g.initializers.pop(name)
    g.nodes.insert(first, Node("Constant", [], [name], {"value": a}, _fresh(g, "Constant")))
    return True


def rw_fp16_init(g, shapes, rng):
    """A weight of a Conv / MatMul stored in fp16 and widened by a Cast (how a half-precision checkpoint exports)."""
    names = sorted({n.inputs[1] for n in g.nodes if n.op in ("Conv", "ConvTranspose", "MatMul") and len(n.inputs) > 1 and n.inputs[1] in g.initializers
                    and g.initializers[n.inputs[1]].dtype == np.float32})
    if not names:
        return False
    name = names[rng.integers(len(names))]
    first = min(k for k, n in enumerate(g.nodes) if name in n.inputs)
    half = _fresh(g, "half")
    g.initializers[half] = g.initializers.pop(name).astype(np.float16)
    g.nodes.insert(first, Node("Cast", [half], [name], {"to": 1}, _fresh(g, "Cast")))
    return True


def rw_reshape_0_m1(g, shapes, rng):
    # (a target computed by a Shape / Gather / Concat chain - what a tracing exporter writes - becomes a constant: the output shape is known)
    sites = [(k, n) for k, n in enumerate(g.nodes) if n.op == "Reshape" and n.inputs[0] in shapes and n.outputs[0] in shapes and not n.attrs.get("allowzero")]
    if not sites:
        return False
    k, n = sites[rng.integers(len(sites))]
    src = shapes[n.inputs[0]]
    full = list(shapes[n.outputs[0]])
    new = list(full)
    zero_ok = [i for i, v in enumerate(full) if i < len(src) and src[i] == v]
    for i in zero_ok:
        if rng.integers(2):
            new[i] = 0
    cand = [i for i, v in enumerate(new) if v != 0]
    if cand and rng.integers(2):
        new[cand[rng.integers(len(cand))]] = -1
    name = _fresh(g, "target")
    g.initializers[name] = np.asarray(new, np.int64)
    n.inputs[1] = name
    return True


def rw_bias_unsqueeze(g, shapes, rng):
    sites = []
    for k, n in enumerate(g.nodes):
        if n.op == "Add":
            for s, i in enumerate(n.inputs):
                if i in g.initializers and g.initializers[i].ndim == 1 and n.inputs[1 - s] in shapes and len(shapes[n.inputs[1 - s]]) >= 3 and not i.startswith("/rw/"):
                    sites.append((k, s, i))
    if not sites:
        return False
    k, s, name = sites[rng.integers(len(sites))]
    a = g.initializers[name]
    out = _fresh(g, "bias")
    if rng.integers(2):        # [N] -> Unsqueeze -> [1, 1, N]
        axes = [0, 1]
        if g.opset >= 13:
            axn = _fresh(g, "axes"); g.initializers[axn] = np.asarray(axes, np.int64)
            node = Node("Unsqueeze", [name, axn], [out], {}, _fresh(g, "Unsqueeze"))
        else:
            node = Node("Unsqueeze", [name], [out], {"axes": axes}, _fresh(g, "Unsqueeze"))
    else:                      # stored [1, N], squeezed back to [N]
        stored = _fresh(g, "bias2d"); g.initializers[stored] = a.reshape(1, -1)
        if g.opset >= 13:
            axn = _fresh(g, "axes"); g.initializers[axn] = np.asarray([0], np.int64)
            node = Node("Squeeze", [stored, axn], [out], {}, _fresh(g, "Squeeze"))
        else:
            node = Node("Squeeze", [stored], [out], {"axes": [0]}, _fresh(g, "Squeeze"))
    g.nodes[k].inputs[s] = out
    g.nodes.insert(k, node)
    return True


def rw_ln_axis(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "LayerNormalization" and n.inputs[0] in shapes]
    if not sites:
        return False
    n = sites[rng.integers(len(sites))]
    r = len(shapes[n.inputs[0]])
    ax = n.attrs.get("axis", -1)
    n.attrs["axis"] = r - 1 if ax < 0 else -1
    return True


def rw_permute(g, shapes, rng):
    """Another topological order of the node list (Kahn's algorithm with a random choice among the ready nodes)."""
    produced_by = {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            if o:
                produced_by[o] = k
    deps = [sorted({produced_by[i] for i in n.inputs if i in produced_by}) for n in g.nodes]
    users = [[] for _ in g.nodes]
    for k, d in enumerate(deps):
        for p in d:
            users[p].append(k)
    left = [len(d) for d in deps]
    ready = [k for k, c in enumerate(left) if c == 0]
    order = []
    while ready:
        k = ready.pop(int(rng.integers(len(ready))))
        order.append(k)
        for u in users[k]:
            left[u] -= 1
            if left[u] == 0:
                ready.append(u)
    assert len(order) == len(g.nodes)
    g.nodes = [g.nodes[k] for k in order]
    return True


def rw_commute(g, shapes, rng):
    sites = [n for n in g.nodes if n.op in ("Add", "Mul") and len(n.inputs) == 2 and (n.inputs[0] in shapes or n.inputs[1] in shapes)]
    if not sites:
        return False
    n = sites[rng.integers(len(sites))]
    n.inputs.reverse()
    return True


def rw_dead(g, shapes, rng):
    names = sorted(k for k in shapes if not k.startswith("__"))
    src = names[rng.integers(len(names))]
    k = max((i for i, n in enumerate(g.nodes) if src in n.outputs), default=-1)
    g.nodes.insert(k + 1, Node("Shape" if rng.integers(2) else "Relu", [src], [_fresh(g, "dead")], {}, _fresh(g, "Dead")))
    return True


def _const_value(g, name):
    """The value of an initializer or of a Constant node's output (None if `name` is neither)."""
    if name in g.initializers:
        return g.initializers[name]
    for n in g.nodes:
        if n.op == "Constant" and n.outputs and n.outputs[0] == name and isinstance(n.attrs.get("value"), np.ndarray):
            return n.attrs["value"]
    return None


def _qkv_sites(g):
    """(index of the [2,0,3,1,4] Transpose, {0: gather node index, 1: .., 2: ..}) of every attention block whose q, k, v are three Gathers on axis 0."""
    sites = []
    for k, n in enumerate(g.nodes):
        if n.op == "Transpose" and list(n.attrs.get("perm", [])) == [2, 0, 3, 1, 4]:
            us = _consumers(g, n.outputs[0])
            if len(us) == 3 and all(u.op == "Gather" and u.attrs.get("axis", 0) == 0 for _, u in us):
                idx = {}
                for ku, u in us:
                    v = _const_value(g, u.inputs[1])
                    if v is not None and v.size == 1:
                        idx[int(v.reshape(-1)[0])] = ku
                if sorted(idx) == [0, 1, 2]:
                    sites.append((k, idx))
    return sites


def rw_split_qkv(g, shapes, rng):
    sites = _qkv_sites(g)
    if not sites:
        return False
    k, idx = sites[rng.integers(len(sites))]
    t = g.nodes[k].outputs[0]
    outs = [g.nodes[idx[i]].outputs[0] for i in range(3)]
    parts = [_fresh(g, "part") for _ in range(3)]
    new = []
    if g.opset >= 13 and rng.integers(2):       # explicit sizes as the second input (opset >= 13) ...
        sz = _fresh(g, "sizes"); g.initializers[sz] = np.asarray([1, 1, 1], np.int64)
        new.append(Node("Split", [t, sz], parts, {"axis": 0}, _fresh(g, "Split")))
    elif g.opset < 13 and rng.integers(2):      # ... or as the attribute (opset < 13) ...
        new.append(Node("Split", [t], parts, {"axis": 0, "split": [1, 1, 1]}, _fresh(g, "Split")))
    else:                                       # ... or equal parts, one per output
        new.append(Node("Split", [t], parts, {"axis": 0}, _fresh(g, "Split")))
    for pn, on in zip(parts, outs):
        if on in shapes:
            shapes[pn] = (1,) + tuple(shapes[on])
        if g.opset >= 13:
            axn = _fresh(g, "axes"); g.initializers[axn] = np.asarray([0], np.int64)
            new.append(Node("Squeeze", [pn, axn], [on], {}, _fresh(g, "Squeeze")))
        else:
            new.append(Node("Squeeze", [pn], [on], {"axes": [0]}, _fresh(g, "Squeeze")))
    first = min(idx.values())
    for ku in sorted(idx.values(), reverse=True):
        del g.nodes[ku]
    g.nodes[first:first] = new
    return True


def rw_sdpa_scale(g, shapes, rng):
    sites = []
    for k, idx in _qkv_sites(g):
        q, kk = g.nodes[idx[0]].outputs[0], g.nodes[idx[1]].outputs[0]
        qm = [(i, n) for i, n in _consumers(g, q) if n.op == "Mul"]
        kt = [(i, n) for i, n in _consumers(g, kk) if n.op == "Transpose" and list(n.attrs.get("perm", [])) == [0, 1, 3, 2]]
        if len(qm) == 1 and len(kt) == 1 and len(_consumers(g, q)) == 1 and len(_consumers(g, kk)) == 1:
            other = [i for i in qm[0][1].inputs if i != q]
            c = _const_value(g, other[0]) if len(other) == 1 else None
            if c is None and len(other) == 1:
                c = shapes.get("__static_scalars__", {}).get(other[0])
            if c is not None and c.size == 1 and c.dtype == np.float32 and float(c.reshape(-1)[0]) > 0 and not other[0].startswith("/rw/"):
                sites.append((qm[0][0], kt[0][0], other[0], float(c.reshape(-1)[0])))
    if not sites:
        return False
    iq, ikt, cname, s = sites[rng.integers(len(sites))]
    r = _fresh(g, "sqrt_scale")
    g.initializers[r] = np.asarray(np.sqrt(np.float32(s)), np.float32)
    qmul, ktr = g.nodes[iq], g.nodes[ikt]
    qmul.inputs = [r if i == cname else i for i in qmul.inputs]
    if rng.integers(2):      # k^T * sqrt(s) behind the transpose ...
        raw = _fresh(g, "kT")
        out = ktr.outputs[0]
        ktr.outputs = [raw]
        if out in shapes:
            shapes[raw] = shapes[out]
        g.nodes.insert(ikt + 1, Node("Mul", [raw, r], [out], {}, _fresh(g, "Mul")))
    else:                    # ... or k * sqrt(s) in front of it
        src = ktr.inputs[0]
        scaled = _fresh(g, "k_scaled")
        if src in shapes:
            shapes[scaled] = shapes[src]
        ktr.inputs = [scaled]
        g.nodes.insert(ikt, Node("Mul", [src, r], [scaled], {}, _fresh(g, "Mul")))
    return True


def rw_ln_decompose(g, shapes, rng):
    sites = [(k, n) for k, n in enumerate(g.nodes) if n.op == "LayerNormalization" and n.inputs[0] in shapes and n.attrs.get("axis", -1) in (-1, len(shapes[n.inputs[0]]) - 1)
             and len(n.outputs) == 1 and len(n.inputs) >= 2]
    if not sites:
        return False
    k, n = sites[rng.integers(len(sites))]
    x, gamma = n.inputs[0], n.inputs[1]
    beta = n.inputs[2] if len(n.inputs) > 2 and n.inputs[2] else None
    eps = _fresh(g, "eps"); g.initializers[eps] = np.asarray(n.attrs.get("epsilon", 1e-5), np.float32)
    m, d, sq, var, ve, sd, nrm, sc = (_fresh(g, s) for s in ("mean", "centred", "squared", "var", "var_eps", "std", "normed", "scaled"))
    sh = shapes[x]
    for name in (d, sq, nrm, sc):
        shapes[name] = sh
    for name in (m, var, ve, sd):
        shapes[name] = tuple(sh[:-1]) + (1,)
    stem = n.name or "LayerNormalization"
    new = [Node("ReduceMean", [x], [m], {"axes": [-1], "keepdims": 1}, stem + "/ReduceMean"), Node("Sub", [x, m], [d], {}, stem + "/Sub")]
    if rng.integers(2):
        two = _fresh(g, "two"); g.initializers[two] = np.asarray(2.0, np.float32)
        new.append(Node("Pow", [d, two], [sq], {}, stem + "/Pow"))
    else:
        new.append(Node("Mul", [d, d], [sq], {}, stem + "/Mul_sq"))
    new += [Node("ReduceMean", [sq], [var], {"axes": [-1], "keepdims": 1}, stem + "/ReduceMean_1"), Node("Add", [var, eps], [ve], {}, stem + "/Add"),
            Node("Sqrt", [ve], [sd], {}, stem + "/Sqrt"), Node("Div", [d, sd], [nrm], {}, stem + "/Div")]
    out = n.outputs[0]
    new.append(Node("Mul", [nrm, gamma], [sc if beta else out], {}, stem + "/Mul"))
    if beta:
        new.append(Node("Add", [sc, beta], [out], {}, stem + "/Add_1"))
    g.nodes[k:k + 1] = new
    return True


def rw_gelu_op(g, shapes, rng):
    prod = {n.outputs[0]: n for n in g.nodes}
    sites = []
    for e in g.nodes:
        if e.op != "Erf":
            continue
        d = prod.get(e.inputs[0])
        if d is None or d.op != "Div" or d.inputs[0] not in shapes:
            continue
        c = _const_value(g, d.inputs[1])
        ua = _consumers(g, e.outputs[0])
        if c is None or c.size != 1 or abs(float(c.reshape(-1)[0]) - 2 ** 0.5) > 1e-6 or len(ua) != 1 or ua[0][1].op != "Add" or len(_consumers(g, d.outputs[0])) != 1:
            continue
        a = ua[0][1]
        one = [_const_value(g, i) for i in a.inputs if i != e.outputs[0]]
        um = _consumers(g, a.outputs[0])
        if len(one) != 1 or one[0] is None or one[0].size != 1 or float(one[0].reshape(-1)[0]) != 1.0 or len(um) != 1 or um[0][1].op != "Mul" or d.inputs[0] not in um[0][1].inputs:
            continue
        m = um[0][1]
        uh = _consumers(g, m.outputs[0])
        if len(uh) != 1 or uh[0][1].op != "Mul":
            continue
        hm = uh[0][1]
        half = [_const_value(g, i) for i in hm.inputs if i != m.outputs[0]]
        if len(half) != 1 or half[0] is None or half[0].size != 1 or float(half[0].reshape(-1)[0]) != 0.5:
            continue
        if sorted(k for k, n in _consumers(g, d.inputs[0])) != sorted([g.nodes.index(d), g.nodes.index(m)]):
            continue
        sites.append((d, e, a, m, hm))
    if not sites:
        return False
    d, e, a, m, hm = sites[rng.integers(len(sites))]
    k = g.nodes.index(d)
    node = Node("Gelu", [d.inputs[0]], [hm.outputs[0]], {"approximate": "none"} if rng.integers(2) else {}, hm.name or _fresh(g, "Gelu"))
    g.nodes = [n for n in g.nodes if n not in (d, e, a, m, hm)]
    g.nodes.insert(k, node)
    g.opset = max(g.opset, 20)
    return True


def rw_d2s_dcr(g, shapes, rng):
    prod = {n.outputs[0]: n for n in g.nodes}
    sites = []
    for d in g.nodes:
        if d.op != "DepthToSpace" or d.attrs.get("mode", "DCR") != "CRD":
            continue
        t = prod.get(d.inputs[0])
        a = prod.get(t.inputs[0]) if t is not None and t.op == "Transpose" and list(t.attrs.get("perm", [])) == [0, 3, 1, 2] else None
        if a is None or a.op != "Add" or len(_consumers(g, a.outputs[0])) != 1 or len(_consumers(g, t.outputs[0])) != 1:
            continue
        bname = [i for i in a.inputs if i in g.initializers]
        mm = [prod.get(i) for i in a.inputs if i not in g.initializers]
        if len(bname) != 1 or len(mm) != 1 or mm[0] is None or mm[0].op != "MatMul" or mm[0].inputs[1] not in g.initializers or len(_consumers(g, mm[0].outputs[0])) != 1:
            continue
        if len(_consumers(g, mm[0].inputs[1])) != 1 or len(_consumers(g, bname[0])) != 1:
            continue
        sites.append((d, a, mm[0], bname[0]))
    if not sites:
        return False
    d, a, mm, bname = sites[rng.integers(len(sites))]
    r = int(d.attrs["blocksize"]); rr = r * r
    w, b = g.initializers[mm.inputs[1]], g.initializers[bname]
    N = w.shape[1]; oc = N // rr
    # DCR column s * oc + c holds what CRD column c * rr + s held
    src = np.asarray([c * rr + s_ for s_ in range(rr) for c in range(oc)])
    wn, bn = _fresh(g, "w_dcr"), _fresh(g, "b_dcr")
    g.initializers[wn] = np.ascontiguousarray(w[:, src]); g.initializers[bn] = np.ascontiguousarray(b[src])
    mm.inputs[1] = wn
    a.inputs = [bn if i == bname else i for i in a.inputs]
    if rng.integers(2):
        d.attrs["mode"] = "DCR"
    else:
        d.attrs.pop("mode")
    return True


REWRITES = {"gemm": rw_gemm, "identity": rw_identity, "dropout": rw_dropout, "cast": rw_cast, "transpose2": rw_transpose2, "squeeze": rw_squeeze,
            "const_node": rw_const_node, "fp16_init": rw_fp16_init, "reshape_0_m1": rw_reshape_0_m1, "bias_unsqueeze": rw_bias_unsqueeze,
            "ln_axis": rw_ln_axis, "permute": rw_permute, "commute": rw_commute, "dead": rw_dead, "split_qkv": rw_split_qkv, "ln_decompose": rw_ln_decompose,
            "sdpa_scale": rw_sdpa_scale, "gelu_op": rw_gelu_op, "d2s_dcr": rw_d2s_dcr}
INEXACT = ("fp16_init", "sdpa_scale")                      # weights rounded to fp16 / sqrt(s)^2 for s: the plan keeps its text, not its bytes
EXACT = [k for k in REWRITES if k not in INEXACT]          # rewrites under which the engine file must not change by a byte


def rewrite(g: Graph, shapes: dict, seed: int, kinds=None, count=None):
    """A deep copy of g with `count` (default 2-5) rewrites drawn from `kinds` (default: all) applied at random sites.  .applied lists what took."""
    rng = np.random.default_rng(seed)
    v = copy.copy(g)                                   # nodes are copied, weight arrays shared (no rewrite writes into an array)
    v.nodes = [Node(n.op, list(n.inputs), list(n.outputs), dict(n.attrs), n.name) for n in g.nodes]
    v.initializers = dict(g.initializers)
    v.inputs, v.outputs = list(g.inputs), list(g.outputs)
    shapes = dict(shapes)
    kinds = list(kinds or REWRITES)
    v.applied = []
    for _ in range(count or int(rng.integers(2, 6))):
        kind = kinds[rng.integers(len(kinds))]
        if kind == "permute" and "permute" in v.applied:
            continue
        if REWRITES[kind](v, shapes, rng):
            v.applied.append(kind)
    return v


if __name__ == "__main__":      # tools/onnx_rewrite.py in.onnx out.onnx batch tile seed [kind ...]
    src, dst, batch, tile, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    gg = onnx_reader.load(src)
    vv = rewrite(gg, runtime_shapes(src, batch, tile), seed, sys.argv[6:] or None)
    dump(vv, dst, packed=bool(seed & 1))
    print(f"{dst}: {', '.join(vv.applied)}")
