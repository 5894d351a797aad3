"""Writes tests/golden/plan_text.json: sha256 of the op lines of describe_plan() for the two headline graph families as tools/synth_models.py exports
them (seed 7, batch 2, tile 64).  Pinned in round 5, when the graph simplifier (csrc/simplify.cpp) went in front of the lowering: the texts were
compared with round 4's loader (its host sources from git cd47354 built with g++) and were identical line for line - the canonical form of a graph
the exporter wrote is that graph.  Re-run only when the plan format or the lowering changes on purpose."""
import hashlib
import importlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth_models as sm  # noqa: E402

pkg = importlib.import_module("waifu2x-tensorrt_amd")
out = {}
with tempfile.TemporaryDirectory() as d:
    for family, (model, scale, batch, tile) in {"cunet_s2": ("cunet/art", 2, 2, 64), "swin_unet_s4": ("swin_unet/art", 4, 2, 64)}.items():
        p = os.path.join(d, family + ".onnx")
        sm.export_onnx(sm.make_model(model, scale, seed=7), p, batch=batch, tile=tile)
        out[family] = hashlib.sha256("\n".join(pkg.describe_plan(p, batch, tile).splitlines()[2:]).encode()).hexdigest()
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "plan_text.json"), "w"), indent=1)
print(out)
