"""MI355X-native waifu2x engine: Python host mirror of the reference's trt::Img2Img interface.

The product is `libw2x.so` (HIP kernels + C++ engine, built from csrc/ by the Makefile or
`__graft_entry__.build()`); this package only binds its C ABI (include/w2x/c_api.h) with ctypes.
There is no CPU fallback: constructing an engine without the shared library raises.
"""
from .engine import (BuildConfig, RenderConfig, Img2Img, Precision, Severity, lib, lib_path,
                     calculate_tiles, strip_plan, shard_plan, render_sharded, device_pci_bus_id, ipc_open, ipc_close, tile_weights, describe_plan, write_engine_file, validate_engine_file, sha256_hex, W2xError,
                     debug_switches)

__all__ = ["BuildConfig", "RenderConfig", "Img2Img", "Precision", "Severity", "lib", "lib_path",
           "calculate_tiles", "strip_plan", "shard_plan", "render_sharded", "device_pci_bus_id", "ipc_open", "ipc_close", "tile_weights", "describe_plan", "write_engine_file", "validate_engine_file", "sha256_hex", "W2xError", "debug_switches"]
