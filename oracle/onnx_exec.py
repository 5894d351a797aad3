"""ORACLE (test infrastructure, not product code): fp32 CPU executor for ONNX graphs.

Stands in for what the reference delegates to TensorRT: IExecutionContext::enqueueV3
(src/tensorrt/img2img_infer.cpp:80) on an engine built from the ONNX file
(src/tensorrt/img2img_build.cpp:54-173).  TensorRT is closed source and absent, so this
follows the ONNX operator specification (opset 13-17) node by node, in fp32, NCHW -
"parity unpinned" against TensorRT itself (no TensorRT, no golden outputs in the reference);
it is cross-checked against the torch modules the fixtures were exported from
(tests/test_oracle_net.py).

Two arithmetic modes:
  * fp32 (default): every value in float32 - the ONNX specification's reading of the graph;
  * act_dtype="float16": the model of a kFP16 engine (img2img_build.cpp:128) - weights rounded to fp16, products
    accumulated in fp32, and every value a layer hands to the next layer rounded to fp16: the output of a
    Conv / ConvTranspose / MatMul / Gemm (after the constant bias Add that follows it, if any), of an activation,
    LayerNormalization, Softmax, Clip, and of an Add / Mul of two runtime tensors (residuals, gates).  It makes the
    checker's output fp16-grained like the engine's, so that differences can be read in fp16 ULPs; the exact
    places where TensorRT rounds are unknown (closed source), this is the layer-by-layer reading.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import onnx_reader


def _t(a):
    if isinstance(a, torch.Tensor):
        return a
    a = np.asarray(a)
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(a.shape)


_CAST = {1: torch.float32, 6: torch.int32, 7: torch.int64, 9: torch.bool, 10: torch.float16,
         11: torch.float64, 2: torch.uint8, 3: torch.int8}


class Executor:
    _HEAVY = {"Conv", "ConvTranspose", "MatMul", "Gemm"}
    _ROUND_ALWAYS = {"LeakyRelu", "Relu", "Sigmoid", "LayerNormalization", "Softmax", "Clip", "Tanh"}

    def __init__(self, path_or_graph, threads: int | None = None, act_dtype: str = "float32"):
        g = onnx_reader.load(path_or_graph) if isinstance(path_or_graph, str) else path_or_graph
        self.g = g
        self.consts = {k: _t(v) for k, v in g.initializers.items()}
        self.threads = threads
        self._static = None  # values that do not depend on the graph input (folded once)
        assert act_dtype in ("float32", "float16")
        self.half = act_dtype == "float16"
        if self.half:
            self._plan_rounding()

    def _plan_rounding(self):
        """Which node outputs are rounded to fp16 (see the module docstring) and which constants are fp16 weights."""
        g = self.g
        users, producer = {}, {}
        for k, n in enumerate(g.nodes):
            for i in n.inputs:
                if i: users.setdefault(i, []).append(k)
            for o in n.outputs:
                producer[o] = k
        # runtime values = everything reachable from the graph input
        runtime = {g.inputs[0].name}
        for n in g.nodes:
            if any(i in runtime for i in n.inputs if i):
                runtime.update(n.outputs)
        self._round = set()
        for k, n in enumerate(g.nodes):
            if not n.outputs or n.outputs[0] not in runtime:
                continue
            out = n.outputs[0]
            rt_in = [i for i in n.inputs if i and i in runtime]
            if n.op in self._HEAVY:
                if n.op in ("MatMul", "Gemm") and len(rt_in) == 2:
                    self._round.add(k)                     # q k^T and attn v: both operands are activations
                    continue
                us = users.get(out, [])
                bias_next = len(us) == 1 and g.nodes[us[0]].op == "Add" and sum(1 for i in g.nodes[us[0]].inputs if i in runtime) == 1
                if not bias_next:
                    self._round.add(k)
            elif n.op in self._ROUND_ALWAYS:
                self._round.add(k)
            elif n.op == "Add":
                if len(rt_in) == 2 or any(producer.get(i) is not None and g.nodes[producer[i]].op in self._HEAVY for i in rt_in):
                    self._round.add(k)
            elif n.op == "Mul" and len(rt_in) == 2:
                self._round.add(k)
        # weights of the heavy nodes are what the engine stores in fp16
        for n in g.nodes:
            if n.op in self._HEAVY and len(n.inputs) > 1 and n.inputs[1] in self.consts and self.consts[n.inputs[1]].is_floating_point():
                self.consts[n.inputs[1]] = self.consts[n.inputs[1]].half().float()

    # -- public -----------------------------------------------------------------------------
    @property
    def input_name(self):
        return self.g.inputs[0].name

    def run(self, x: np.ndarray, keep: tuple = ()) -> np.ndarray | dict:
        """x: [B,3,T,T] float32 in [0,1] -> [B,3,T',T'] float32."""
        if self.threads:
            torch.set_num_threads(self.threads)
        env = dict(self.consts)
        env[self.input_name] = _t(np.asarray(x, np.float32))
        if self.half:
            env[self.input_name] = env[self.input_name].half().float()
        # values are dropped after their last reader: a swin_unet graph has ~700 runtime nodes and keeping every
        # intermediate alive costs tens to hundreds of GB at tile 256..640
        last = {}
        for k, n in enumerate(self.g.nodes):
            for i in n.inputs:
                if i: last[i] = k
        pinned = set(self.consts) | {o.name for o in self.g.outputs} | set(keep)
        with torch.no_grad():
            for k, n in enumerate(self.g.nodes):
                outs = self._node(n, [env[i] if i else None for i in n.inputs])
                if not isinstance(outs, (tuple, list)):
                    outs = (outs,)
                if self.half and k in self._round and outs[0].dtype == torch.float32:
                    outs = (outs[0].half().float(),) + tuple(outs[1:])
                for name, v in zip(n.outputs, outs):
                    env[name] = v
                for i in n.inputs:
                    if i and last.get(i) == k and i not in pinned:
                        env.pop(i, None)
        y = env[self.g.outputs[0].name].numpy()
        if keep:
            return {k: env[k].numpy() for k in keep} | {"y": y}
        return y

    # -- ops --------------------------------------------------------------------------------
    def _node(self, n, a):
        op, at = n.op, n.attrs
        if op == "Constant":
            v = at.get("value")
            if v is None:
                if "value_float" in at: v = np.asarray(at["value_float"], np.float32)
                elif "value_int" in at: v = np.asarray(at["value_int"], np.int64)
                elif "value_ints" in at: v = np.asarray(at["value_ints"], np.int64)
                elif "value_floats" in at: v = np.asarray(at["value_floats"], np.float32)
            return _t(v)
        if op == "Conv":
            pads = at.get("pads", [0, 0, 0, 0])
            assert pads[0] == pads[2] and pads[1] == pads[3]
            return F.conv2d(a[0], a[1], a[2] if len(a) > 2 else None, stride=at.get("strides", [1, 1]),
                            padding=(pads[0], pads[1]), dilation=at.get("dilations", [1, 1]),
                            groups=at.get("group", 1))
        if op == "ConvTranspose":
            pads = at.get("pads", [0, 0, 0, 0])
            assert pads[0] == pads[2] and pads[1] == pads[3]
            return F.conv_transpose2d(a[0], a[1], a[2] if len(a) > 2 else None,
                                      stride=at.get("strides", [1, 1]), padding=(pads[0], pads[1]),
                                      output_padding=at.get("output_padding", [0, 0]),
                                      groups=at.get("group", 1), dilation=at.get("dilations", [1, 1]))
        if op == "LeakyRelu":
            return F.leaky_relu(a[0], at.get("alpha", 0.01))
        if op == "Relu":
            return F.relu(a[0])
        if op == "Sigmoid":
            return torch.sigmoid(a[0])
        if op == "Erf":
            return torch.erf(a[0])
        if op == "Gelu":                                  # opset 20; approximate = "none" (erf) | "tanh"
            return F.gelu(a[0], approximate=at.get("approximate", "none"))
        if op == "Sqrt":
            return torch.sqrt(a[0])
        if op == "Exp":
            return torch.exp(a[0])
        if op == "Tanh":
            return torch.tanh(a[0])
        if op == "Neg":
            return -a[0]
        if op == "Not":
            return ~a[0]
        if op == "Add":
            return a[0] + a[1]
        if op == "Sub":
            return a[0] - a[1]
        if op == "Mul":
            return a[0] * a[1]
        if op == "Div":
            if not a[0].is_floating_point() and not a[1].is_floating_point():
                return torch.div(a[0], a[1], rounding_mode="trunc")
            return a[0] / a[1]
        if op == "Mod":      # ONNX Mod: fmod = 0 -> the sign of the divisor (Python / torch.remainder), fmod = 1 -> the sign of the dividend (C fmod)
            return torch.fmod(a[0], a[1]) if at.get("fmod", 0) else torch.remainder(a[0], a[1])
        if op == "Pow":
            return torch.pow(a[0], a[1])
        if op == "Equal":
            return a[0] == a[1]
        if op == "Less":
            return a[0] < a[1]
        if op == "Greater":
            return a[0] > a[1]
        if op == "Where":
            return torch.where(a[0], a[1], a[2])
        if op == "Clip":
            lo = a[1] if len(a) > 1 and a[1] is not None else at.get("min")
            hi = a[2] if len(a) > 2 and a[2] is not None else at.get("max")
            y = a[0]
            if lo is not None: y = torch.maximum(y, torch.as_tensor(lo, dtype=y.dtype))
            if hi is not None: y = torch.minimum(y, torch.as_tensor(hi, dtype=y.dtype))
            return y
        if op == "MatMul":
            return torch.matmul(a[0], a[1])
        if op == "Gemm":
            A = a[0].T if at.get("transA", 0) else a[0]
            B = a[1].T if at.get("transB", 0) else a[1]
            y = at.get("alpha", 1.0) * (A @ B)
            if len(a) > 2 and a[2] is not None:
                y = y + at.get("beta", 1.0) * a[2]
            return y
        if op == "Softmax":
            return torch.softmax(a[0], dim=at.get("axis", -1))
        if op == "LayerNormalization":
            ax = at.get("axis", -1)
            shape = a[0].shape[ax:] if ax < 0 else a[0].shape[ax:]
            return F.layer_norm(a[0], tuple(shape), a[1], a[2] if len(a) > 2 else None, at.get("epsilon", 1e-5))
        if op == "ReduceMean":
            axes = at.get("axes")
            if axes is None and len(a) > 1:
                axes = a[1].tolist()
            return torch.mean(a[0], dim=tuple(axes), keepdim=bool(at.get("keepdims", 1)))
        if op == "GlobalAveragePool":
            return torch.mean(a[0], dim=(2, 3), keepdim=True)
        if op == "Shape":
            return torch.tensor(list(a[0].shape), dtype=torch.int64)
        if op == "Cast":
            return a[0].to(_CAST[at["to"]])
        if op == "Reshape":
            shp = a[1].tolist()
            shp = [a[0].shape[i] if s == 0 else s for i, s in enumerate(shp)]
            return a[0].reshape(shp)
        if op == "Flatten":
            ax = at.get("axis", 1)
            return a[0].reshape(int(np.prod(a[0].shape[:ax])), -1)
        if op == "Transpose":
            perm = at.get("perm") or list(range(a[0].dim()))[::-1]
            return a[0].permute(perm).contiguous()
        if op == "Unsqueeze":
            axes = at.get("axes") if "axes" in at else a[1].tolist()
            y = a[0]
            nd = y.dim() + len(axes)
            for ax in sorted(x % nd for x in axes):
                y = y.unsqueeze(ax)
            return y
        if op == "Squeeze":
            axes = at.get("axes") if "axes" in at else (a[1].tolist() if len(a) > 1 else None)
            if axes is None:
                return a[0].squeeze()
            y = a[0]
            for ax in sorted((x % y.dim() for x in axes), reverse=True):
                y = y.squeeze(ax)
            return y
        if op == "Concat":
            return torch.cat([t for t in a], dim=at["axis"])
        if op == "Split":       # several outputs: qkv.unbind(0) exports as Split + Squeeze
            ax = at.get("axis", 0)
            sizes = at.get("split") or (a[1].tolist() if len(a) > 1 and a[1] is not None else None)
            return list(torch.split(a[0], sizes if sizes else a[0].shape[ax] // len(n.outputs), dim=ax))
        if op == "Slice":
            starts, ends = a[1].tolist(), a[2].tolist()
            axes = a[3].tolist() if len(a) > 3 and a[3] is not None else list(range(len(starts)))
            steps = a[4].tolist() if len(a) > 4 and a[4] is not None else [1] * len(starts)
            y = a[0]
            for s, e, ax, st in zip(starts, ends, axes, steps):
                d = y.shape[ax]
                if st > 0:
                    s = max(0, min(d, s + d if s < 0 else s))
                    e = max(0, min(d, e + d if e < 0 else e))
                    idx = torch.arange(s, e, st)
                else:
                    s = max(-1, min(d - 1, s + d if s < 0 else s))
                    e = max(-1, min(d - 1, e + d if e < 0 else e)) if e > -(1 << 62) else -1
                    idx = torch.arange(s, e, st)
                y = y.index_select(ax, idx)
            return y
        if op == "Gather":
            ax = at.get("axis", 0)
            idx = a[1]
            idx = torch.where(idx < 0, idx + a[0].shape[ax], idx)
            if idx.dim() == 0:
                return a[0].select(ax, int(idx))
            y = a[0].index_select(ax, idx.reshape(-1))
            shp = list(a[0].shape[:ax]) + list(idx.shape) + list(a[0].shape[ax + 1:])
            return y.reshape(shp)
        if op == "ConstantOfShape":
            v = at.get("value")
            v = _t(v).reshape(()) if v is not None else torch.tensor(0.0)
            return torch.full(a[0].tolist(), v.item(), dtype=v.dtype)
        if op == "Expand":
            shp = a[1].tolist()
            return a[0] * torch.ones(shp, dtype=a[0].dtype) if a[0].dtype != torch.bool else \
                (a[0].to(torch.int8) * torch.ones(shp, dtype=torch.int8)).bool()
        if op == "Range":
            return torch.arange(a[0].item(), a[1].item(), a[2].item(), dtype=a[0].dtype)
        if op == "ScatterND":
            y = a[0].clone()
            idx, upd = a[1], a[2]
            k = idx.shape[-1]
            flat_idx = idx.reshape(-1, k)
            upd = upd.reshape((flat_idx.shape[0],) + tuple(y.shape[k:]))
            y[tuple(flat_idx[:, j] for j in range(k))] = upd
            return y
        if op == "Pad":
            pads = a[1].tolist() if len(a) > 1 else at["pads"]
            mode = at.get("mode", "constant")
            val = float(a[2]) if len(a) > 2 and a[2] is not None and a[2].numel() else 0.0
            nd = a[0].dim()
            tp = []
            for d in range(nd - 1, -1, -1):
                tp += [pads[d], pads[d + nd]]
            if mode == "constant":
                return F.pad(a[0], tp, value=val)
            # replicate/reflect only on the last two dims
            return F.pad(a[0], tp[:4], mode={"edge": "replicate", "reflect": "reflect"}[mode])
        if op == "DepthToSpace":
            b, c, h, w = a[0].shape
            r = at["blocksize"]
            if at.get("mode", "DCR") == "CRD":
                return F.pixel_shuffle(a[0], r)
            t = a[0].reshape(b, r, r, c // (r * r), h, w).permute(0, 3, 4, 1, 5, 2)
            return t.reshape(b, c // (r * r), h * r, w * r)
        if op in ("Identity", "Dropout"):      # (Dropout at inference is the identity)
            return a[0]
        raise NotImplementedError(f"oracle: ONNX op {op}")


def count_flops(path_or_graph, x_shape) -> dict:
    """Algorithmic FLOPs = sum over Conv/ConvTranspose/MatMul/Gemm of 2*MACs at the static shape
    (SURVEY.md section 8d definition), obtained by running the graph once on zeros."""
    ex = Executor(path_or_graph)
    total = {"Conv": 0, "ConvTranspose": 0, "MatMul": 0, "Gemm": 0}
    orig = ex._node

    def wrapped(n, a):
        out = orig(n, a)
        if n.op == "Conv":
            w = a[1]
            total["Conv"] += 2 * out.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        elif n.op == "ConvTranspose":
            w = a[1]
            total["ConvTranspose"] += 2 * a[0].numel() * w.shape[1] * w.shape[2] * w.shape[3]
        elif n.op == "MatMul" and a[0].is_floating_point():
            total["MatMul"] += 2 * out.numel() * a[0].shape[-1]
        elif n.op == "Gemm":
            total["Gemm"] += 2 * out.numel() * a[0].shape[-1]
        return out

    ex._node = wrapped
    ex.run(np.zeros(x_shape, np.float32))
    total["total"] = sum(total.values())
    return total
