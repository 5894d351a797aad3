"""The oracle's ONNX reader/executor against the torch modules the synthetic graphs were exported from
(an independent second opinion for the whole-graph output: SURVEY.md section 8c (iv))."""
import numpy as np
import pytest
import torch

import synth_models as sm
from oracle import onnx_exec, onnx_reader


@pytest.mark.parametrize("model,scale,tile,small", [("cunet/art", 2, 64, False), ("cunet/art", 1, 64, False),
                                                     ("swin_unet/art", 4, 64, True), ("swin_unet/art", 2, 40, True),
                                                     ("swin_unet/art", 1, 40, True)])
def test_executor_matches_torch_module(onnx_model, model, scale, tile, small):
    path = onnx_model(model, scale, 2, tile, small=small)
    net = sm.make_model(model, scale, seed=1234 + 3, small=small)
    x = torch.rand(2, 3, tile, tile, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = net(x).numpy()
    y = onnx_exec.Executor(path).run(x.numpy())
    assert y.shape == ref.shape == (2, 3, sm.output_tile_size(model, scale, tile), sm.output_tile_size(model, scale, tile))
    assert np.abs(y - ref).max() < 2e-5


def test_reader_sees_expected_structure(onnx_model):
    g = onnx_reader.load(onnx_model("swin_unet/art", 4, 1, 64, small=True))
    assert g.opset == 17 and len(g.inputs) == 1 and g.inputs[0].shape == ["b", 3, 64, 64]
    ops = {n.op for n in g.nodes}
    assert {"Conv", "MatMul", "Softmax", "LayerNormalization", "Erf", "DepthToSpace", "Clip"} <= ops
    assert g.initializers["patch.0.weight"].shape == (24, 3, 3, 3)


def test_flop_count_matches_survey_estimate(onnx_model):
    # cunet s2 ~2.4 GF/tile @T=64
    f = onnx_exec.count_flops(onnx_model("cunet/art", 2, 1, 64), (1, 3, 64, 64))
    assert f["total"] == 2390222336


def test_model_path_naming():
    # main.cpp:201-204 incl. the trailing underscore for scale 1 (quirk Q7)
    assert sm.model_path("r", "swin_unet/art", 4, 3).endswith("models/swin_unet/art/noise3_scale4x.onnx")
    assert sm.model_path("r", "cunet/art", 1, 0).endswith("models/cunet/art/noise0_.onnx")
    assert sm.model_path("r", "cunet/art", 2, -1).endswith("models/cunet/art/scale2x.onnx")
