import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run under gpurun)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes binding of libw2x.so); builds the library if it is missing."""
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "waifu2x-tensorrt_amd", "libw2x.so")):
        g.build()
    return g.package()


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("w2x_models"))


@pytest.fixture(scope="session")
def onnx_model(model_dir):
    """Factory: export (and cache) a synthetic-weight graph -> path of models/<model>/<name>.onnx."""
    import synth_models as sm
    cache = {}

    def get(model, scale, batch, tile, noise=3, small=False, opset=17, variant=None, dynamic=True):
        vkey = tuple(sorted((variant or {}).items()))
        key = (model, scale, batch, tile, noise, small, opset, vkey, dynamic)
        if key not in cache:
            tag = "".join(f"_{k}{v}" for k, v in vkey) + ("" if dynamic else "_static")
            root = os.path.join(model_dir, f"b{batch}_t{tile}_{'s' if small else 'f'}_o{opset}{tag}")
            path = sm.model_path(root, model, scale, noise)
            sm.export_onnx(sm.make_model(model, scale, seed=1234 + noise, small=small, variant=variant), path, batch, tile, opset=opset, dynamic=dynamic)
            cache[key] = path
        return cache[key]

    return get
