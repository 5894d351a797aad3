"""Test infrastructure: SEMANTIC mutations of an ONNX graph - each one changes the function the graph computes.

tools/onnx_rewrite.py asks "does every spelling of the same function lower to the same plan?".  This file asks the opposite question: does the
loader READ what it lowers, or does it recognise a shape and fill in what it expects?  The reference hands any ONNX file to TensorRT's parser
(src/tensorrt/img2img_build.cpp:81-88), which executes what the file says; a lowering that pattern-matches a Swin block and hard-codes the LayerNorm
epsilon, the LeakyRelu slope, the attention scale, the roll distance or the operand order of q and k would pass every test that is built from
tools/synth_models.py's own exports and be wrong on the first file that differs.  Every mutation below changes one such quantity at one site:

    leaky_alpha       LeakyRelu alpha -> another slope
    leaky_to_relu     LeakyRelu -> Relu
    clip_bounds       Clip(0, 1) -> Clip(lo, hi)
    ln_eps            LayerNormalization epsilon x 100 .. x 10000
    attn_scale        the scalar the attention scores (q, k or q k^T) are multiplied by -> x 0.5 .. x 2
    div_scale         q * s  ->  s / q   (a Div with the constant as the dividend: not a scale)
    transpose_weight  a square MatMul weight W -> W^T; a Conv weight with Cin == Cout -> in / out channels swapped
    roll_shift        the distance of one torch.roll (Slice + Slice + Concat) -> another distance
    drop_residual     y = a + b (two runtime tensors)  ->  y = b
    d2s_mode          DepthToSpace CRD <-> DCR with the producer's columns left as they are (the loader lowers both modes: the mutant must be followed)
    conv_drop_bias    a Conv / ConvTranspose loses its bias input
    matmul_drop_bias  the Add(bias) behind a MatMul goes
    gelu_const        one of the constants of the erf GELU chain (1 / sqrt 2, the + 1, the 0.5) -> another value
    swap_qk           the Gather indices that pick q and k out of the packed qkv tensor swapped
    softmax_axis      Softmax axis -1 -> -2
    bias_table        the relative-position bias table of one block transposed over (query, key)
    se_gate           cunet squeeze-excite: Sigmoid -> Relu on the gate

`mutate(g, shapes, seed)` applies ONE mutation at a random site and returns the variant (`.applied` = [kind], `.site` = node name).  The caller decides
whether the mutation took effect (oracle output of the variant differs from the original's) - e.g. a transposed symmetric matrix changes nothing.
What the loader owes a mutant: a plan that computes the MUTANT (checked on the GPU against the oracle run on the mutant,
tests/test_gpu_parity.py::test_mutated_graphs_follow_the_oracle) or a refusal naming the node; on the CPU: never the engine file of the original
(tests/test_loader_mutations.py)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import onnx_rewrite as rw  # noqa: E402
from oracle.onnx_reader import Node  # noqa: E402


def _new_const(g, stem, value):
    name = rw._fresh(g, stem)
    g.initializers[name] = np.asarray(value)
    return name


def _pick(rng, seq):
    return seq[int(rng.integers(len(seq)))]


def _scalar(g, name):
    c = rw._const_value(g, name)
    return c if c is not None and c.size == 1 else None


def mu_leaky_alpha(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "LeakyRelu"]
    if not sites:
        return None
    n = _pick(rng, sites)
    cur = float(n.attrs.get("alpha", 0.01))
    n.attrs["alpha"] = float(_pick(rng, [a for a in (0.02, 0.05, 0.2, 0.3) if abs(a - cur) > 1e-3]))
    return n.name


def mu_leaky_to_relu(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "LeakyRelu"]
    if not sites:
        return None
    n = _pick(rng, sites)
    n.op = "Relu"; n.attrs = {}
    return n.name


def mu_clip_bounds(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "Clip" and len(n.inputs) >= 3]
    if not sites:
        return None
    n = _pick(rng, sites)
    lo, hi = _pick(rng, [(0.1, 0.9), (0.0, 0.8), (0.2, 1.0), (-0.5, 0.5)])
    n.inputs = [n.inputs[0], _new_const(g, "clip_lo", np.float32(lo)), _new_const(g, "clip_hi", np.float32(hi))]
    return n.name


def mu_ln_eps(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "LayerNormalization"]
    if not sites:
        return None
    n = _pick(rng, sites)
    n.attrs["epsilon"] = float(n.attrs.get("epsilon", 1e-5)) * float(_pick(rng, [100.0, 1000.0, 10000.0]))
    return n.name


def _scale_sites(g):
    """(Mul node, index of its scalar operand) for the multiplication that scales q (torch: q * head_dim ** -0.5) in every attention block"""
    out = []
    for _, idx in rw._qkv_sites(g):
        q = g.nodes[idx[0]].outputs[0]
        for _, m in rw._consumers(g, q):
            if m.op == "Mul":
                other = [k for k, i in enumerate(m.inputs) if i != q]
                if len(other) == 1:
                    out.append((m, other[0]))
    return out


def mu_attn_scale(g, shapes, rng):
    sites = _scale_sites(g)
    if not sites:
        return None
    m, k = _pick(rng, sites)
    c = _scalar(g, m.inputs[k])
    if c is None:
        c = shapes.get("__static_scalars__", {}).get(m.inputs[k])
    if c is None:
        return None
    m.inputs[k] = _new_const(g, "scale", np.float32(float(np.asarray(c).reshape(-1)[0]) * float(_pick(rng, [0.5, 0.7, 1.4, 2.0]))))
    return m.name


def mu_div_scale(g, shapes, rng):
    sites = _scale_sites(g)
    if not sites:
        return None
    m, k = _pick(rng, sites)
    c = _scalar(g, m.inputs[k])
    if c is None:
        c = shapes.get("__static_scalars__", {}).get(m.inputs[k])
    if c is None:
        return None
    s = _new_const(g, "dividend", np.float32(np.asarray(c).reshape(-1)[0]))
    q = m.inputs[1 - k]
    m.op = "Div"; m.inputs = [s, q]          # s / q
    return m.name


def mu_transpose_weight(g, shapes, rng):
    sites = []
    for n in g.nodes:
        if n.op == "MatMul" and n.inputs[1] in g.initializers and g.initializers[n.inputs[1]].ndim == 2 and g.initializers[n.inputs[1]].shape[0] == g.initializers[n.inputs[1]].shape[1]:
            sites.append((n, 1, (1, 0)))
        if n.op == "Conv" and n.inputs[1] in g.initializers and g.initializers[n.inputs[1]].ndim == 4 and g.initializers[n.inputs[1]].shape[0] == g.initializers[n.inputs[1]].shape[1] \
                and int(n.attrs.get("group", 1)) == 1:
            sites.append((n, 1, (1, 0, 2, 3)))
    if not sites:
        return None
    n, k, perm = _pick(rng, sites)
    n.inputs[k] = _new_const(g, "wT", np.ascontiguousarray(g.initializers[n.inputs[k]].transpose(perm)))
    return n.name


def _roll_sites(g, shapes):
    """Concat(Slice(x, [s], [big], [axis]), Slice(x, [0], [s], [axis])) - what torch.roll(x, -s, axis) exports to (and its inverse with s = size - shift)"""
    out = []
    prod = {n.outputs[0]: n for n in g.nodes if n.op == "Slice"}
    consts = {n.outputs[0]: n.attrs["value"] for n in g.nodes if n.op == "Constant" and n.outputs and isinstance(n.attrs.get("value"), np.ndarray)}
    cv = lambda name: g.initializers.get(name, consts.get(name))
    for c in g.nodes:
        if c.op != "Concat" or len(c.inputs) != 2:
            continue
        a, b = prod.get(c.inputs[0]), prod.get(c.inputs[1])
        if a is None or b is None or a.inputs[0] != b.inputs[0] or len(a.inputs) < 4 or len(b.inputs) < 4:
            continue
        va = [cv(i) for i in a.inputs[1:4]]
        vb = [cv(i) for i in b.inputs[1:4]]
        if any(v is None or v.size != 1 for v in va + vb):
            continue
        sa, ea, xa = (int(v.reshape(-1)[0]) for v in va)
        sb, eb, xb = (int(v.reshape(-1)[0]) for v in vb)
        if xa == xb == int(c.attrs.get("axis", 0)) and sb == 0 and eb == sa and ea > 1 << 20 and a.inputs[0] in shapes:
            out.append((c, a, b, sa, xa))
    return out


def mu_roll_shift(g, shapes, rng):
    sites = _roll_sites(g, shapes)
    if not sites:
        return None
    c, a, b, s, ax = _pick(rng, sites)
    size = shapes[a.inputs[0]][ax]
    # the cut point is taken as the file writes it: positive for a forward roll, negative (from the end) for the roll back
    cands = [v for v in ((1, 2, 4, 5) if s >= 0 else (-1, -2, -4, -5)) if v != s and abs(v) < size]
    if not cands:
        return None
    s2 = int(_pick(rng, cands))
    a.inputs[1] = _new_const(g, "roll_start", np.asarray([s2], np.int64))
    b.inputs[2] = _new_const(g, "roll_end", np.asarray([s2], np.int64))
    return c.name


def mu_drop_residual(g, shapes, rng):
    sites = []
    for n in g.nodes:
        if n.op == "Add" and all(i in shapes and i not in g.initializers for i in n.inputs) and shapes[n.inputs[0]] == shapes[n.inputs[1]] and len(shapes[n.inputs[0]]) >= 3:
            sites.append(n)
    if not sites:
        return None
    n = _pick(rng, sites)
    keep = n.inputs[int(rng.integers(2))]
    n.op = "Identity"; n.inputs = [keep]; n.attrs = {}
    return n.name


def mu_d2s_mode(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "DepthToSpace"]
    if not sites:
        return None
    n = _pick(rng, sites)
    n.attrs["mode"] = "DCR" if n.attrs.get("mode", "DCR") == "CRD" else "CRD"
    return n.name


def mu_conv_drop_bias(g, shapes, rng):
    sites = [n for n in g.nodes if n.op in ("Conv", "ConvTranspose") and len(n.inputs) == 3 and n.inputs[2]]
    if not sites:
        return None
    n = _pick(rng, sites)
    n.inputs = n.inputs[:2]
    return n.name


def mu_matmul_drop_bias(g, shapes, rng):
    sites = []
    for n in g.nodes:
        if n.op == "MatMul" and n.inputs[1] in g.initializers:
            us = rw._consumers(g, n.outputs[0])
            if len(us) == 1 and us[0][1].op == "Add":
                other = [i for i in us[0][1].inputs if i != n.outputs[0]]
                if len(other) == 1 and other[0] in g.initializers and g.initializers[other[0]].ndim == 1:
                    sites.append(us[0][1])
    if not sites:
        return None
    a = _pick(rng, sites)
    keep = [i for i in a.inputs if i not in g.initializers][0]
    a.op = "Identity"; a.inputs = [keep]; a.attrs = {}
    return a.name


def mu_gelu_const(g, shapes, rng):
    """x * 0.5 * (1 + erf(x / sqrt 2)) as the exporter writes it: Div(x, c0) -> Erf -> Add(c1) -> Mul(x, .) -> Mul(c2)"""
    sites = []
    prod = {n.outputs[0]: n for n in g.nodes}
    for e in g.nodes:
        if e.op != "Erf":
            continue
        d = prod.get(e.inputs[0])
        us = rw._consumers(g, e.outputs[0])
        if d is None or d.op not in ("Div", "Mul") or len(us) != 1 or us[0][1].op != "Add":
            continue
        a = us[0][1]
        cand = [(d, [k for k, i in enumerate(d.inputs) if _scalar(g, i) is not None]), (a, [k for k, i in enumerate(a.inputs) if _scalar(g, i) is not None])]
        u2 = rw._consumers(g, a.outputs[0])
        if len(u2) == 1 and u2[0][1].op == "Mul":
            u3 = rw._consumers(g, u2[0][1].outputs[0])
            if len(u3) == 1 and u3[0][1].op == "Mul":
                m = u3[0][1]
                cand.append((m, [k for k, i in enumerate(m.inputs) if _scalar(g, i) is not None]))
        sites += [(n, ks[0]) for n, ks in cand if len(ks) == 1]
    if not sites:
        return None
    n, k = _pick(rng, sites)
    c = _scalar(g, n.inputs[k])
    n.inputs[k] = _new_const(g, "gelu_c", np.asarray(float(c.reshape(-1)[0]) * float(_pick(rng, [0.6, 0.8, 1.25, 1.5])), c.dtype).reshape(c.shape))
    return n.name


def mu_swap_qk(g, shapes, rng):
    sites = rw._qkv_sites(g)
    if not sites:
        return None
    _, idx = _pick(rng, sites)
    nq, nk = g.nodes[idx[0]], g.nodes[idx[1]]
    nq.inputs[1], nk.inputs[1] = nk.inputs[1], nq.inputs[1]
    return nq.name


def mu_softmax_axis(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "Softmax"]
    if not sites:
        return None
    n = _pick(rng, sites)
    n.attrs["axis"] = -2
    return n.name


def mu_bias_table(g, shapes, rng):
    sites = []
    prod = {n.outputs[0]: n for n in g.nodes}
    for n in g.nodes:
        if n.op != "Add":
            continue
        for k, i in enumerate(n.inputs):
            c = g.initializers.get(i)
            o = prod.get(n.inputs[1 - k])
            if c is not None and c.ndim >= 3 and c.shape[-1] == c.shape[-2] and c.shape[-1] > 1 and o is not None and o.op == "MatMul":
                sites.append((n, k))
    if not sites:
        return None
    n, k = _pick(rng, sites)
    n.inputs[k] = _new_const(g, "biasT", np.ascontiguousarray(np.swapaxes(g.initializers[n.inputs[k]], -1, -2)))
    return n.name


def mu_se_gate(g, shapes, rng):
    sites = [n for n in g.nodes if n.op == "Sigmoid"]
    if not sites:
        return None
    n = _pick(rng, sites)
    n.op = "Relu"
    return n.name


MUTATIONS = {"leaky_alpha": mu_leaky_alpha, "leaky_to_relu": mu_leaky_to_relu, "clip_bounds": mu_clip_bounds, "ln_eps": mu_ln_eps, "attn_scale": mu_attn_scale,
             "div_scale": mu_div_scale, "transpose_weight": mu_transpose_weight, "roll_shift": mu_roll_shift, "drop_residual": mu_drop_residual,
             "d2s_mode": mu_d2s_mode, "conv_drop_bias": mu_conv_drop_bias, "matmul_drop_bias": mu_matmul_drop_bias, "gelu_const": mu_gelu_const,
             "swap_qk": mu_swap_qk, "softmax_axis": mu_softmax_axis, "bias_table": mu_bias_table, "se_gate": mu_se_gate}


def mutate(g, shapes: dict, seed: int, kinds=None):
    """A copy of g with ONE mutation (kind and site drawn from the seed; kinds that have no site in this graph are skipped).  .applied = [kind], .site = node name."""
    rng = np.random.default_rng(seed)
    kinds = list(kinds or MUTATIONS)
    order = kinds[seed % len(kinds):] + kinds[:seed % len(kinds)]      # kinds in turn (seed k starts at kind k): a few dozen seeds cover them all; the site is drawn
    for kind in order:
        v = _copy(g)
        site = MUTATIONS[kind](v, dict(shapes), rng)
        if site is not None:
            v.applied = [kind]; v.site = site
            return v
    raise RuntimeError("no mutation has a site in this graph")


def _copy(g):
    import copy
    v = copy.copy(g)
    v.nodes = [Node(n.op, list(n.inputs), list(n.outputs), dict(n.attrs), n.name) for n in g.nodes]
    v.initializers = dict(g.initializers)
    v.inputs, v.outputs = list(g.inputs), list(g.outputs)
    return v


if __name__ == "__main__":      # tools/onnx_mutate.py in.onnx out.onnx batch tile seed [kind ...]
    from oracle import onnx_reader
    src, dst, batch, tile, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    gg = onnx_reader.load(src)
    vv = mutate(gg, rw.runtime_shapes(src, batch, tile), seed, sys.argv[6:] or None)
    rw.dump(vv, dst)
    print(f"{dst}: {vv.applied[0]} at {vv.site}")
