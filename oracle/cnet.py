"""ORACLE (test infrastructure, not product code): ctypes binding of oracle/cnet/onnx_net.cpp, the C++ / OpenMP loop-level
restatement of the network (what the reference hands to TensorRT: img2img_infer.cpp:80, img2img_build.cpp:54-173).

    ex = cnet.Executor(path); y = ex.run(x)          # [B,3,T,T] float32 -> [B,3,T',T'] float32, like onnx_exec.Executor in fp32 mode

The library is built on demand into oracle/_build/ (g++ -O3 -fopenmp; __graft_entry__.build() does it up front).
Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libw2x_oracle_net.so")
_lib = None


def build() -> str:
    src = os.path.join(_HERE, "cnet", "onnx_net.cpp")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(_HERE, "cnet")], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.onet_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]; L.onet_load.restype = C.c_void_p
        L.onet_run.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.onet_run.restype = C.c_longlong
        L.onet_flops.argtypes = [C.c_void_p]; L.onet_flops.restype = C.c_double
        L.onet_threads.argtypes = []; L.onet_threads.restype = C.c_int
        L.onet_free.argtypes = [C.c_void_p]; L.onet_free.restype = None
        _lib = L
    return _lib


class Executor:
    def __init__(self, path: str, threads: int = 0):
        self._L = lib()
        err = C.create_string_buffer(512)
        self._h = self._L.onet_load(os.fsencode(path), err, 512)
        if not self._h:
            raise RuntimeError("C++ oracle: " + err.value.decode(errors="replace"))
        self.threads = threads

    def run(self, x: np.ndarray, out_hw_max: int = 0) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        assert x.ndim == 4
        b, c, h, w = x.shape
        cap = b * max(c, 4) * (h * 4) * (w * 4) if not out_hw_max else b * 4 * out_hw_max * out_hw_max     # these graphs scale by at most 4
        y = np.empty(cap, np.float32)
        shp = (C.c_longlong * 4)()
        err = C.create_string_buffer(1024)
        n = self._L.onet_run(self._h, x.ctypes.data, b, c, h, w, y.ctypes.data, cap, shp, int(self.threads), err, 1024)
        if n < 0:
            raise RuntimeError("C++ oracle: " + err.value.decode(errors="replace"))
        return y[:n].reshape(tuple(int(v) for v in shp)).copy()

    @property
    def flops(self) -> float:
        """2 * MACs of the Conv / ConvTranspose / MatMul / Gemm nodes of the last run (SURVEY 8d's algorithmic FLOPs)"""
        return float(self._L.onet_flops(self._h))

    def close(self):
        if self._h:
            self._L.onet_free(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
