"""The loader against the exporter's OWN switches.

The model files the reference loads are exports somebody else made (README.md:11-15; src/tensorrt/img2img_build.cpp:81-88 hands any file to TensorRT's parser):
which opset, whether constants were folded, whether initializers are listed as inputs, whether the module was in training mode or scripted is their choice.
None of those files is reachable here, but the legacy TorchScript exporter that writes them is, with every switch it has.  Each graph family goes through

    do_constant_folding on / off, keep_initializers_as_inputs, training = PRESERVE and TRAINING, dynamic height / width axes, torch.jit.script instead
    of tracing, opset 9 ... 20

and every file the exporter manages to write must lower to THE PLAN of the default export (op lines compared with node names removed: a scripted module
names its nodes differently) - or be refused naming the node.  What the exporter itself cannot write (swin_unet below opset 11: index_put; a scripted swin
module) is listed, not hidden.  Found this way in round 6: opset 20 writes GELU as one `Gelu` node, which the loader refused ("unsupported operator") - now lowered
(approximate = "none"; "tanh" is refused by name) and read by both oracle executors."""
import re

import numpy as np
import pytest
import torch

import synth_models as sm
from oracle import cnet, onnx_exec

FAMILIES = {"cunet_s2": ("cunet/art", 2), "swin_unet_s4": ("swin_unet/art", 4)}
BATCH, TILE = 2, 64
SWITCHES = [("no_constant_folding", dict(do_constant_folding=False)),
            ("keep_initializers_as_inputs", dict(keep_initializers_as_inputs=True)),
            ("training_preserve", dict(training=torch.onnx.TrainingMode.PRESERVE)),
            ("training_mode", dict(training=torch.onnx.TrainingMode.TRAINING, do_constant_folding=False)),
            ("dynamic_hw", dict(dynamic_axes={"x": {0: "b", 2: "h", 3: "w"}, "y": {0: "b", 2: "h", 3: "w"}})),
            ("scripted", dict(script=True))] + [(f"opset{o}", dict(opset=o)) for o in (9, 10, 11, 12, 13, 14, 15, 16, 18, 19, 20)]
# what the EXPORTER cannot write (checked: the failure must be the exporter's, with these words)
EXPORTER_LIMITS = {("swin_unet_s4", "opset9"): "index_put", ("swin_unet_s4", "opset10"): "index_put", ("swin_unet_s4", "scripted"): ""}


def plan_ops(pkg, path):
    """op lines of the plan text without the node names in brackets and without per-file sizes"""
    lines = pkg.describe_plan(path, BATCH, TILE).splitlines()[2:]
    return "\n".join(re.sub(r"\s*\[[^\]]*\]\s*$", "", l) for l in lines)


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_every_exporter_switch_gives_the_plan_of_the_default_export(pkg, tmp_path, family):
    model, scale = FAMILIES[family]
    ref = str(tmp_path / "default.onnx")
    sm.export_onnx(sm.make_model(model, scale, seed=7), ref, BATCH, TILE)
    want = plan_ops(pkg, ref)
    x = np.random.default_rng(3).random((BATCH, 3, TILE, TILE), dtype=np.float32)
    y_ref = onnx_exec.Executor(ref).run(x)
    same, refused, unexportable = [], [], []
    for name, kw in SWITCHES:
        path = str(tmp_path / f"{name}.onnx")
        try:
            sm.export_onnx(sm.make_model(model, scale, seed=7), path, BATCH, TILE, **kw)
        except Exception as e:                                          # the exporter's own limit: only where it is a known one
            assert (family, name) in EXPORTER_LIMITS and EXPORTER_LIMITS[(family, name)] in str(e), (family, name, str(e)[:300])
            unexportable.append(name)
            continue
        try:
            got = plan_ops(pkg, path)
        except pkg.W2xError as e:
            assert "cannot lower node" in str(e) or "graph:" in str(e) or "fold:" in str(e), (family, name, str(e))
            refused.append((name, str(e)[:120]))
            continue
        assert got == want, (family, name)
        same.append(name)
        if name in ("no_constant_folding", "opset11", "opset13", "opset20", "scripted"):      # the checkers read these spellings too, and they are the same function
            ya, yb = onnx_exec.Executor(path).run(x), cnet.Executor(path).run(x)
            assert float(np.abs(ya - yb).max()) < 2e-5 and float(np.abs(ya - y_ref).max()) < 2e-5, (family, name)
    print(f"{family}: same plan {same}; refused {refused}; the exporter could not write {unexportable}")
    assert not refused, refused
    assert len(same) >= len(SWITCHES) - len([k for k in EXPORTER_LIMITS if k[0] == family])


def test_gelu_operator_of_opset_20(pkg, tmp_path):
    """opset 20's Gelu node: approximate = none is the erf form (same engine file as the erf chain of opset 17); the tanh approximation is another function and is refused."""
    import hashlib
    import onnx_rewrite as rw
    from oracle import onnx_reader
    p17, p20 = str(tmp_path / "o17.onnx"), str(tmp_path / "o20.onnx")
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=7), p17, 1, 64, opset=17)
    sm.export_onnx(sm.make_model("swin_unet/art", 4, seed=7), p20, 1, 64, opset=20)
    g = onnx_reader.load(p20)
    assert sum(n.op == "Gelu" for n in g.nodes) == 14 and not any(n.op == "Erf" for n in g.nodes)
    out = []
    for p in (p17, p20):
        assert pkg.write_engine_file(p, 1, 64, p + ".w2x")
        out.append(hashlib.sha256(open(p + ".w2x", "rb").read()).hexdigest())
    assert out[0] == out[1]
    for n in g.nodes:
        if n.op == "Gelu":
            n.attrs["approximate"] = "tanh"
            break
    pt = str(tmp_path / "tanh.onnx")
    rw.dump(g, pt)
    with pytest.raises(pkg.W2xError) as e:
        pkg.describe_plan(pt, 1, 64)
    assert "cannot lower node Gelu" in str(e.value) and "tanh" in str(e.value)
    x = np.random.default_rng(1).random((1, 3, 64, 64), dtype=np.float32)
    assert float(np.abs(onnx_exec.Executor(pt).run(x) - cnet.Executor(pt).run(x)).max()) < 2e-5       # both checkers read the tanh form alike
