"""The loader against graphs it was not handed by tools/synth_models.py.

The reference gives ANY ONNX file to TensorRT's parser (src/tensorrt/img2img_build.cpp:81-88; the files themselves are release assets,
README.md:11-15, unreachable offline).  tools/onnx_rewrite.py re-spells each exported graph the ways exporters, opsets and optimiser passes
spell the same computation (Gemm for MatMul + Add inside a 2-D sandwich, Identity / Dropout / no-op Cast / Transpose pairs / Unsqueeze-Squeeze
and Flatten-Reshape pairs on edges, initializers as Constant nodes, fp16-stored weights, Reshape targets with 0 and -1, biases behind
Unsqueeze / Squeeze, LayerNormalization axis -1 <-> rank - 1 or decomposed into its ReduceMean chain, q / k / v through Split + Squeeze instead of three
Gathers, the attention scale split over q and k^T the way a decomposed scaled_dot_product_attention writes it, the erf GELU chain as opset 20's one Gelu node,
swapped Add / Mul operands, dead nodes, any
topological node order, packed and unpacked repeated fields).  For every variant, seeded:

  * the loader (csrc/fold.cpp -> simplify.cpp -> lower.cpp) must write the ENGINE FILE OF THE ORIGINAL, byte for byte (plan text for the variants
    that store weights in fp16, whose weight bytes legitimately differ) - or refuse naming a node; it must never produce another plan;
  * on a sample of the variants both oracle executors (torch operators / C++ loops) run the variant and agree with each other and with the
    original graph's output: the rewrites preserve the function, and the checkers read the new spellings too.

Default: 200 variants per graph family; W2X_REWRITE_VARIANTS=N overrides.  The GPU part (ten variants rendered, frames equal to the original's) is
tests/test_gpu_parity.py::test_rewritten_graphs_render_the_bytes_of_the_original."""
import collections
import hashlib
import os

import numpy as np
import pytest

import onnx_rewrite as rw
import synth_models as sm
from oracle import cnet, onnx_exec, onnx_reader

FAMILIES = {   # name: (model, scale, batch, tile)
    "cunet_s2": ("cunet/art", 2, 2, 64),
    "cunet_s1": ("cunet/art", 1, 2, 64),
    "swin_unet_s4": ("swin_unet/art", 4, 2, 64),
    "swin_unet_s2": ("swin_unet/art", 2, 1, 64),
}
N_VARIANTS = int(os.environ.get("W2X_REWRITE_VARIANTS", "200"))
EXEC_EVERY = 25        # every 25th variant also runs through both oracle executors


def engine_sha(pkg, onnx_path, batch, tile, out):
    if not pkg.write_engine_file(onnx_path, batch, tile, out):
        return None
    return hashlib.sha256(open(out, "rb").read()).hexdigest()


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_every_spelling_of_a_graph_lowers_to_the_plan_of_the_original(pkg, tmp_path, family):
    model, scale, batch, tile = FAMILIES[family]
    path = str(tmp_path / "m.onnx")
    sm.export_onnx(sm.make_model(model, scale, seed=7), path, batch=batch, tile=tile)
    g = onnx_reader.load(path)
    shapes = rw.runtime_shapes(path, batch, tile)
    ref_sha = engine_sha(pkg, path, batch, tile, str(tmp_path / "ref.w2x"))
    ref_txt = pkg.describe_plan(path, batch, tile)
    assert ref_sha
    x = np.random.default_rng(3).random((batch, 3, tile, tile), dtype=np.float32)
    y_ref = onnx_exec.Executor(path).run(x)
    vpath, vplan = str(tmp_path / "v.onnx"), str(tmp_path / "v.w2x")
    seen, refused, ran = collections.Counter(), [], 0
    for seed in range(N_VARIANTS):
        v = rw.rewrite(g, shapes, seed)
        rw.dump(v, vpath, packed=bool(seed & 1))
        seen.update(set(v.applied))
        tag = f"{family} seed {seed}: {', '.join(v.applied)}"
        if set(v.applied) & set(rw.INEXACT):
            try:
                assert pkg.describe_plan(vpath, batch, tile) == ref_txt, tag
            except pkg.W2xError as e:
                refused.append((tag, str(e)))
        else:
            sha = engine_sha(pkg, vpath, batch, tile, vplan)
            if sha is None:
                try:
                    pkg.describe_plan(vpath, batch, tile)
                    raise AssertionError(tag + ": the engine file could not be written but the plan lowers")
                except pkg.W2xError as e:
                    refused.append((tag, str(e)))
            else:
                assert sha == ref_sha, tag + ": another plan:\n" + pkg.describe_plan(vpath, batch, tile)[:2000]
        if seed % EXEC_EVERY == 0:
            ya, yb = onnx_exec.Executor(vpath).run(x), cnet.Executor(vpath).run(x)
            tol = 2e-3 if "fp16_init" in v.applied else 2e-5          # (weights rounded to fp16 move the output by ~1e-4; sqrt(s)^2 for s by ~1e-7)
            assert float(np.abs(ya - yb).max()) < 2e-5, tag + ": the two oracle executors disagree"
            assert float(np.abs(ya - y_ref).max()) < tol, tag + ": the rewrite changed the function"
            ran += 1
    # a refusal must name the node it stops at ("cannot lower node <op> \"<name>\": why"); none is expected from these rewrites
    for tag, msg in refused:
        assert "cannot lower node" in msg or "graph:" in msg or "fold:" in msg, (tag, msg)
    assert not refused, refused[:5]
    if N_VARIANTS >= 100:
        want = set(rw.REWRITES) - ({"gemm", "ln_axis", "bias_unsqueeze", "split_qkv", "ln_decompose", "sdpa_scale", "reshape_0_m1", "gelu_op", "d2s_dcr"} if family.startswith("cunet") else set())   # (no such sites in a cunet graph, or a single one)
        assert want <= set(seen), (sorted(want - set(seen)), dict(seen))      # every kind of rewrite took part
    print(f"{family}: {N_VARIANTS} variants, {ran} through both oracle executors; rewrites applied: {dict(sorted(seen.items()))}")


def test_simplifier_leaves_the_exporters_own_graphs_alone(pkg, tmp_path):
    """The canonical form of a graph the exporter wrote is that graph: same ops in the same order as round 4's loader produced (the op lines of the
    plan text of the two headline families, pinned by their sha256 in tests/golden/plan_text.json)."""
    import json
    pinned = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plan_text.json")))
    for family in ("cunet_s2", "swin_unet_s4"):
        model, scale, batch, tile = FAMILIES[family]
        path = str(tmp_path / f"{family}.onnx")
        sm.export_onnx(sm.make_model(model, scale, seed=7), path, batch=batch, tile=tile)
        txt = pkg.describe_plan(path, batch, tile)
        ops = "\n".join(l for l in txt.splitlines()[2:])
        assert hashlib.sha256(ops.encode()).hexdigest() == pinned[family], family
