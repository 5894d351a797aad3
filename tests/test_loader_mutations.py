"""Does the loader READ what it lowers?  (CPU half; the GPU half is tests/test_gpu_parity.py::test_mutated_graphs_follow_the_oracle.)

The reference gives any ONNX file to TensorRT's parser, which executes what the file says (src/tensorrt/img2img_build.cpp:81-88).  This loader pattern-matches
whole Swin blocks and convolution groups onto fused operators; a matcher that recognises the shape of a block and fills in the LayerNorm epsilon, the LeakyRelu
slope, the attention scale, the roll distance or the order of q and k it EXPECTS would pass every test built from tools/synth_models.py's own exports.
tools/onnx_mutate.py changes exactly one such quantity at one site per variant (17 kinds).  For every seeded mutant:

  * the loader refuses, naming a node - or it writes an engine file, and that file is NOT the engine file of the original (a lowering that ignored the mutated
    quantity writes the original's bytes; where the two files agree the oracle must say the mutation changed nothing, e.g. a transposed symmetric matrix);
  * it never crashes and never reports a failure without a node.

What the engine file of a mutant computes is checked where it can run: ten mutants per family are built, run through w2x_infer on the GPU and compared with
the oracle executing the MUTANT within the parity bounds of the original.

Default: 100 mutants for the two headline families, 34 (two of every kind) for the other two (W2X_MUTANTS=N overrides both)."""
import collections
import hashlib
import os

import numpy as np
import pytest

import onnx_mutate as om
import onnx_rewrite as rw
import synth_models as sm
from oracle import onnx_exec, onnx_reader

FAMILIES = {   # name: (model, scale, batch, tile, mutants)
    "cunet_s2": ("cunet/art", 2, 2, 64, 100),
    "cunet_s1": ("cunet/art", 1, 2, 64, 34),
    "swin_unet_s4": ("swin_unet/art", 4, 2, 64, 100),
    "swin_unet_s2": ("swin_unet/art", 2, 1, 64, 34),
}
NAMED = ("cannot lower node", "graph:", "fold:")     # every refusal starts from one of these and carries a node's op and name


def sha_of(pkg, onnx_path, batch, tile, out):
    if not pkg.write_engine_file(onnx_path, batch, tile, out):
        return None
    return hashlib.sha256(open(out, "rb").read()).hexdigest()


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_a_mutated_graph_never_lowers_to_the_plan_of_the_original(pkg, tmp_path, family):
    model, scale, batch, tile, count = FAMILIES[family]
    count = int(os.environ.get("W2X_MUTANTS", count))
    path = str(tmp_path / "m.onnx")
    sm.export_onnx(sm.make_model(model, scale, seed=7), path, batch=batch, tile=tile)
    g, shapes = onnx_reader.load(path), rw.runtime_shapes(path, batch, tile)
    ref_sha = sha_of(pkg, path, batch, tile, str(tmp_path / "ref.w2x"))
    assert ref_sha
    x = np.random.default_rng(3).random((batch, 3, tile, tile), dtype=np.float32)
    y_ref = None
    vpath, vplan = str(tmp_path / "v.onnx"), str(tmp_path / "v.w2x")
    built, refused, inert = collections.Counter(), collections.Counter(), []
    for seed in range(count):
        v = om.mutate(g, shapes, seed)
        kind = v.applied[0]
        tag = f"{family} seed {seed}: {kind} at {v.site}"
        rw.dump(v, vpath, packed=bool(seed & 1))
        sha = sha_of(pkg, vpath, batch, tile, vplan)
        if sha is None:
            try:
                pkg.describe_plan(vpath, batch, tile)
                raise AssertionError(tag + ": the engine file could not be written but the plan lowers")
            except pkg.W2xError as e:
                assert any(k in str(e) for k in NAMED) and '"' in str(e), (tag, str(e))
                refused[kind] += 1
            continue
        built[kind] += 1
        if sha == ref_sha:
            # the original's bytes: legitimate only if the mutant IS the original function
            if y_ref is None:
                y_ref = onnx_exec.Executor(path).run(x)
            delta = float(np.abs(onnx_exec.Executor(vpath).run(x) - y_ref).max())
            assert delta < 1e-6, tag + f": the loader wrote the engine file of the ORIGINAL for a graph whose output differs by {delta:.3g}"
            inert.append(tag)
    kinds_here = {k for k in om.MUTATIONS if k in built or k in refused}
    print(f"{family}: {count} mutants; built {dict(sorted(built.items()))}; refused {dict(sorted(refused.items()))}; inert {len(inert)}")
    if count >= 34:
        want = set(om.MUTATIONS) - ({"ln_eps", "attn_scale", "div_scale", "roll_shift", "d2s_mode", "matmul_drop_bias", "gelu_const", "swap_qk", "softmax_axis", "bias_table"}
                                    if family.startswith("cunet") else {"se_gate"})
        if family == "cunet_s1":
            want -= {"drop_residual"} if "drop_residual" not in kinds_here else set()
        assert want <= kinds_here, sorted(want - kinds_here)


def test_every_mutation_kind_changes_the_function(tmp_path):
    """The mutations are what they claim: one variant of each kind, run by the oracle, differs from the original (otherwise the test above proves nothing)."""
    for model, scale, kinds in (("swin_unet/art", 4, [k for k in om.MUTATIONS if k != "se_gate"]),
                                ("cunet/art", 2, ["leaky_alpha", "leaky_to_relu", "clip_bounds", "transpose_weight", "drop_residual", "conv_drop_bias", "se_gate"])):
        path = str(tmp_path / "m.onnx")
        sm.export_onnx(sm.make_model(model, scale, seed=7), path, batch=1, tile=64)
        g, shapes = onnx_reader.load(path), rw.runtime_shapes(path, 1, 64)
        x = np.random.default_rng(3).random((1, 3, 64, 64), dtype=np.float32)
        y_ref = onnx_exec.Executor(path).run(x)
        for kind in kinds:
            v = om.mutate(g, shapes, 1, [kind])
            rw.dump(v, str(tmp_path / "v.onnx"))
            delta = float(np.abs(onnx_exec.Executor(str(tmp_path / "v.onnx")).run(x) - y_ref).max())
            assert delta > 1e-5, (model, kind, v.site, delta)


def test_advisor_cases_of_round_5(pkg, tmp_path):
    """Two graphs the round-5 review named: (1) a Div with the constant as the DIVIDEND where the attention scale sits (c / q is not a scale of q: it was
    folded as scale *= 1 / c); (2) the 2-D sandwich Reshape([M, K]) -> MatMul -> Add(const) -> Reshape whose Add constant is an [M, N] table - legal in the
    2-D form, a different broadcast (or none) once the rows have their leading dimensions back.  Both must be refused or lowered as written, never as the original."""
    path = str(tmp_path / "m.onnx")
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=7), path, batch=1, tile=64)
    g, shapes = onnx_reader.load(path), rw.runtime_shapes(path, 1, 64)
    ref_sha = sha_of(pkg, path, 1, 64, str(tmp_path / "ref.w2x"))
    # (1)
    v = om.mutate(g, shapes, 0, ["div_scale"])
    rw.dump(v, str(tmp_path / "d.onnx"))
    with pytest.raises(pkg.W2xError) as e:
        pkg.describe_plan(str(tmp_path / "d.onnx"), 1, 64)
    assert "cannot lower node" in str(e.value)
    # (2) a sandwich around fc1 whose bias is stored as a full [M, N] table
    v = rw.rewrite(g, shapes, 0, kinds=["identity"], count=1)      # (a plain copy with one harmless Identity)
    mm = next(n for n in v.nodes if n.op == "MatMul" and n.name.endswith("fc1/MatMul"))
    add = next(n for n in v.nodes if n.op == "Add" and mm.outputs[0] in n.inputs)
    x_name, w_name = mm.inputs
    K, N = v.initializers[w_name].shape
    M = int(np.prod(shapes[x_name][:-1]))
    bias_name = [i for i in add.inputs if i != mm.outputs[0]][0]
    table = np.ascontiguousarray(np.broadcast_to(v.initializers[bias_name], (M, N))).astype(np.float32)
    table[M // 2:] += 0.25                                          # rows of the second half get another bias: no [N] vector says the same
    v.initializers["t2d"] = table
    v.initializers["s2d"] = np.asarray([-1, K], np.int64)
    v.initializers["sNd"] = np.asarray(list(shapes[x_name][:-1]) + [N], np.int64)
    k = v.nodes.index(mm)
    out = add.outputs[0]
    new = [onnx_reader.Node("Reshape", [x_name, "s2d"], ["x2d"], {}, "sandwich/in"), onnx_reader.Node("MatMul", ["x2d", w_name], ["y2d"], {}, mm.name),
           onnx_reader.Node("Add", ["y2d", "t2d"], ["z2d"], {}, add.name), onnx_reader.Node("Reshape", ["z2d", "sNd"], [out], {}, "sandwich/out")]
    v.nodes = [n for n in v.nodes if n is not mm and n is not add]
    v.nodes[k:k] = new
    rw.dump(v, str(tmp_path / "s.onnx"))
    y_ref = onnx_exec.Executor(path).run(np.full((1, 3, 64, 64), 0.5, np.float32))
    y_var = onnx_exec.Executor(str(tmp_path / "s.onnx")).run(np.full((1, 3, 64, 64), 0.5, np.float32))
    assert float(np.abs(y_var - y_ref).max()) > 1e-4             # the variant is another function
    sha = sha_of(pkg, str(tmp_path / "s.onnx"), 1, 64, str(tmp_path / "s.w2x"))
    if sha is None:
        with pytest.raises(pkg.W2xError) as e:
            pkg.describe_plan(str(tmp_path / "s.onnx"), 1, 64)
        assert any(k in str(e.value) for k in NAMED)
    else:
        assert sha != ref_sha


def test_tools_shell_scripts_parse():
    """bash -n over every script under tools/ (round 5 shipped one that a search-and-replace had broken)."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "**", "*.sh"), recursive=True))
    assert scripts
    for s in scripts:
        r = subprocess.run(["bash", "-n", s], capture_output=True, text=True)
        assert r.returncode == 0, (s, r.stderr)
