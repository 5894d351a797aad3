"""The two restatements of the network held against each other: oracle/onnx_exec.py states every ONNX node with
torch-CPU operators, oracle/cnet/onnx_net.cpp states it as plain C++ loops (own protobuf reader, own Conv / MatMul / Softmax /
LayerNormalization ...).  They share nothing but the ONNX file; on every graph family, operator set and export variant the parity
tests use they must give the same tile to fp32 rounding - and both agree with the torch module the file was exported from.  What
TensorRT itself computes stays unpinned (no TensorRT here, no golden outputs in the reference); what this pins is that the checker
does not hang on one implementation's reading of the graph."""
import os

import numpy as np
import pytest
import torch

import synth_models as sm
from oracle import cnet, onnx_exec

TOL = 2e-5       # values in [0, 1]; two fp32 summation orders over up to 1728 products


@pytest.mark.parametrize("model,scale,batch,tile,kw", [
    ("cunet/art", 2, 2, 64, {}), ("cunet/art", 1, 1, 64, {}),
    ("swin_unet/art", 4, 2, 64, dict(small=True)), ("swin_unet/art", 2, 1, 40, dict(small=True)), ("swin_unet/art", 1, 1, 40, dict(small=True)),
    ("swin_unet/art", 4, 1, 64, {}), ("swin_unet/photo", 2, 1, 88, {}), ("swin_unet/art_scan", 1, 1, 64, {}),
    # the export variants of test_loader_takes_graphs_it_was_not_written_around: other operator sets for the same arithmetic
    ("swin_unet/art", 4, 1, 80, dict(variant={"ws": 8})), ("swin_unet/art", 4, 1, 64, dict(variant={"heads": 3})),
    ("swin_unet/art", 4, 2, 64, dict(dynamic=False)), ("swin_unet/art", 4, 1, 64, dict(opset=11)), ("swin_unet/art", 4, 1, 64, dict(opset=13)),
    ("swin_unet/art", 4, 1, 64, dict(variant={"tv": 1}))])
def test_cpp_loops_and_torch_operators_agree(onnx_model, model, scale, batch, tile, kw):
    path = onnx_model(model, scale, batch, tile, **kw)
    x = np.random.default_rng(7).random((batch, 3, tile, tile), dtype=np.float32)
    ref = onnx_exec.Executor(path).run(x)
    ex = cnet.Executor(path)
    y = ex.run(x)
    assert y.shape == ref.shape and y.dtype == np.float32
    d = float(np.abs(y - ref).max())
    assert d < TOL, d
    assert np.array_equal(ex.run(x), y)                                    # second run: folded constants, same bytes
    assert ex.flops == onnx_exec.count_flops(path, x.shape)["total"]       # SURVEY 8d's algorithmic FLOPs, counted by two programs
    one = cnet.Executor(path, threads=1).run(x)
    assert np.array_equal(one, y)                                          # thread count does not change a sum's order
    ex.close()


def test_cpp_oracle_matches_the_torch_module(onnx_model):
    path = onnx_model("swin_unet/art", 4, 2, 64, small=True)
    net = sm.make_model("swin_unet/art", 4, seed=1234 + 3, small=True)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = net(x).numpy()
    assert np.abs(cnet.Executor(path).run(x.numpy()) - ref).max() < TOL


def test_cpp_oracle_refuses_what_it_cannot_read(tmp_path, onnx_model):
    with pytest.raises(RuntimeError, match="cannot open"):
        cnet.Executor(str(tmp_path / "missing.onnx"))
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x3a\xff\xff\xff\x0f" + b"\x00" * 16)
    with pytest.raises(RuntimeError, match="truncated"):
        cnet.Executor(str(bad))
    raw = open(onnx_model("cunet/art", 2, 2, 64), "rb").read()
    cut = tmp_path / "cut.onnx"
    cut.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(RuntimeError):
        cnet.Executor(str(cut))
    ex = cnet.Executor(onnx_model("cunet/art", 2, 2, 64))
    with pytest.raises(RuntimeError, match="Conv"):
        ex.run(np.zeros((1, 5, 64, 64), np.float32))                      # channel count the first convolution does not take
