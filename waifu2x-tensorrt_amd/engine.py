"""ctypes binding of libw2x.so mirroring trt::Img2Img (/root/reference/src/tensorrt/img2img.h:14-50):
same five methods (build, load, render, setMessageCallback, setProgressCallback), same config
structs (config.h:12-43) and the same bool-return + message-callback error convention."""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "libw2x.so")


class W2xError(RuntimeError):
    pass


class Precision(enum.IntEnum):      # config.h:7-10
    TF32 = 0                        # fp32 maps, split-bf16 products (include/w2x/config.h)
    FP16 = 1
    FP32 = 2                        # not in the reference: the same engine with exact fp32 products


class Severity(enum.IntEnum):       # logger.h:11-18
    critical = 0
    error = 1
    warn = 2
    info = 3
    debug = 4
    trace = 5


class _BuildConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "deviceId", "precision", "minBatchSize", "optBatchSize", "maxBatchSize", "minChannels", "optChannels",
        "maxChannels", "minWidth", "optWidth", "maxWidth", "minHeight", "optHeight", "maxHeight")]


class _RenderConfig(C.Structure):
    _fields_ = [("deviceId", C.c_int), ("precision", C.c_int), ("batchSize", C.c_int), ("channels", C.c_int),
                ("height", C.c_int), ("width", C.c_int), ("scaling", C.c_int), ("overlapX", C.c_double),
                ("overlapY", C.c_double), ("tta", C.c_int), ("ttaBugCompat", C.c_int)]


@dataclass
class BuildConfig:                  # defaults of config.h:12-31
    deviceId: int = 0
    precision: Precision = Precision.FP16
    minBatchSize: int = 1
    optBatchSize: int = 1
    maxBatchSize: int = 4
    minChannels: int = 3
    optChannels: int = 3
    maxChannels: int = 3
    minWidth: int = 64
    optWidth: int = 256
    maxWidth: int = 640
    minHeight: int = 64
    optHeight: int = 256
    maxHeight: int = 640

    @staticmethod
    def fixed(batch: int, tile: int, device: int = 0, precision: Precision = Precision.FP16) -> "BuildConfig":
        """min = opt = max, the way the CLI fills it (main.cpp:276-291)."""
        return BuildConfig(device, precision, batch, batch, batch, 3, 3, 3, tile, tile, tile, tile, tile, tile)


@dataclass
class RenderConfig:                 # defaults of config.h:33-43
    deviceId: int = 0
    precision: Precision = Precision.FP16
    batchSize: int = 1
    channels: int = 3
    height: int = 256
    width: int = 256
    scaling: int = 4
    overlap: tuple = (0.0625, 0.0625)
    tta: bool = False
    ttaBugCompat: bool = False


_MSG_FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
_PROG_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_double, C.c_void_p)
_lib = None


def lib():
    """Load libw2x.so and declare every symbol of include/w2x/c_api.h.  Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(lib_path):
        raise W2xError(f"{lib_path} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C waifu2x-tensorrt_amd` (there is no CPU fallback)")
    L = C.CDLL(lib_path)
    vp = C.c_void_p
    L.w2x_create.restype = vp
    L.w2x_destroy.argtypes = [vp]
    L.w2x_set_message_callback.argtypes = [vp, _MSG_FN, vp]
    L.w2x_set_progress_callback.argtypes = [vp, _PROG_FN, vp]
    L.w2x_build.argtypes = [vp, C.c_char_p, C.POINTER(_BuildConfig)]; L.w2x_build.restype = C.c_int
    L.w2x_load.argtypes = [vp, C.c_char_p, C.POINTER(_RenderConfig)]; L.w2x_load.restype = C.c_int
    L.w2x_render.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]; L.w2x_render.restype = C.c_int
    L.w2x_render16.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]; L.w2x_render16.restype = C.c_int
    L.w2x_render_strip.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int]; L.w2x_render_strip.restype = C.c_int
    L.w2x_render_sequence.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, C.c_int]; L.w2x_render_sequence.restype = C.c_int
    L.w2x_alloc_host.argtypes = [vp, C.c_size_t]; L.w2x_alloc_host.restype = vp
    L.w2x_free_host.argtypes = [vp, vp]; L.w2x_free_host.restype = None
    L.w2x_pin_host.argtypes = [vp, vp, C.c_size_t]; L.w2x_pin_host.restype = C.c_int
    L.w2x_unpin_host.argtypes = [vp, vp]; L.w2x_unpin_host.restype = None
    L.w2x_strip_plan.argtypes = [C.c_int] * 7 + [C.c_double, C.c_double, C.c_int, C.c_int, vp]; L.w2x_strip_plan.restype = C.c_int
    L.w2x_render_sharded.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]; L.w2x_render_sharded.restype = C.c_int
    L.w2x_shard_compute.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int]; L.w2x_shard_compute.restype = C.c_int
    L.w2x_shard_slab.argtypes = [vp, vp]; L.w2x_shard_slab.restype = vp
    L.w2x_shard_finish.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, vp, vp]; L.w2x_shard_finish.restype = C.c_int
    L.w2x_ipc_export.argtypes = [vp, vp]; L.w2x_ipc_export.restype = C.c_int
    L.w2x_ipc_open.argtypes = [vp, C.c_int]; L.w2x_ipc_open.restype = vp
    L.w2x_ipc_close.argtypes = [vp]; L.w2x_ipc_close.restype = None
    L.w2x_shard_plan.argtypes = [C.c_int] * 7 + [C.c_double, C.c_double, C.c_int, C.c_int, vp]; L.w2x_shard_plan.restype = C.c_int
    L.w2x_infer.argtypes = [vp, vp, vp]; L.w2x_infer.restype = C.c_int
    L.w2x_output_tile_size.argtypes = [vp]; L.w2x_output_tile_size.restype = C.c_int
    L.w2x_pass_tiles.argtypes = [vp]; L.w2x_pass_tiles.restype = C.c_int
    L.w2x_plan_flops.argtypes = [vp]; L.w2x_plan_flops.restype = C.c_double
    L.w2x_last_render_ms.argtypes = [vp]; L.w2x_last_render_ms.restype = C.c_float
    L.w2x_bench_resident.argtypes = [vp, C.c_int]; L.w2x_bench_resident.restype = C.c_float
    L.w2x_profile_frame.argtypes = [vp, vp, C.c_int]; L.w2x_profile_frame.restype = C.c_int
    L.w2x_op_times.argtypes = [vp, vp, C.c_int]; L.w2x_op_times.restype = C.c_int
    L.w2x_calculate_tiles.argtypes = [C.c_int] * 7 + [C.c_double, C.c_double, vp, vp, C.c_int]; L.w2x_calculate_tiles.restype = C.c_int
    L.w2x_tile_weights.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp]; L.w2x_tile_weights.restype = C.c_int
    L.w2x_describe_plan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]; L.w2x_describe_plan.restype = C.c_int
    L.w2x_describe_plan_precision.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]; L.w2x_describe_plan_precision.restype = C.c_int
    L.w2x_write_engine_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]; L.w2x_write_engine_file.restype = C.c_int
    L.w2x_validate_engine_file.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]; L.w2x_validate_engine_file.restype = C.c_int
    L.w2x_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]; L.w2x_device_pci_bus_id.restype = C.c_int
    L.w2x_sha256_hex.argtypes = [vp, C.c_size_t, C.c_char_p]
    L.w2x_version.restype = C.c_char_p
    L.w2x_debug_set.argtypes = [C.c_char_p, C.c_int]; L.w2x_debug_set.restype = C.c_int
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "w2x_create", "w2x_destroy", "w2x_set_message_callback", "w2x_set_progress_callback", "w2x_build", "w2x_load",
    "w2x_render", "w2x_render16", "w2x_infer", "w2x_output_tile_size", "w2x_plan_flops", "w2x_pass_tiles", "w2x_last_render_ms", "w2x_bench_resident", "w2x_profile_frame", "w2x_op_times",
    "w2x_render_strip", "w2x_strip_plan", "w2x_render_sharded", "w2x_shard_plan", "w2x_shard_compute", "w2x_shard_slab", "w2x_shard_finish", "w2x_ipc_export", "w2x_ipc_open", "w2x_ipc_close", "w2x_render_sequence", "w2x_alloc_host", "w2x_free_host", "w2x_pin_host", "w2x_unpin_host", "w2x_calculate_tiles", "w2x_tile_weights", "w2x_describe_plan", "w2x_describe_plan_precision", "w2x_write_engine_file", "w2x_validate_engine_file", "w2x_device_pci_bus_id", "w2x_sha256_hex", "w2x_version", "w2x_debug_set"]


class Img2Img:
    """trt::Img2Img mirror.  cv::Mat <-> numpy uint8 [rows, cols, 3] BGR."""

    def __init__(self):
        self._L = lib()
        self._h = self._L.w2x_create()
        if not self._h:
            raise W2xError("w2x_create failed")
        self.messages: list[tuple[int, str]] = []
        self._user_msg = None
        self._user_prog = None
        self._msg_cb = _MSG_FN(self._on_msg)
        self._prog_cb = _PROG_FN(self._on_prog)
        self._L.w2x_set_message_callback(self._h, self._msg_cb, None)
        self._L.w2x_set_progress_callback(self._h, self._prog_cb, None)

    def _on_msg(self, sev, msg, _):
        m = msg.decode(errors="replace")
        self.messages.append((sev, m))
        if self._user_msg:
            self._user_msg(Severity(sev), m)

    def _on_prog(self, cur, total, speed, _):
        if self._user_prog:
            self._user_prog(cur, total, speed)

    def close(self):
        if self._h:
            self._L.w2x_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setMessageCallback(self, cb):
        self._user_msg = cb

    def setProgressCallback(self, cb):
        self._user_prog = cb

    def last_error(self) -> str:
        errs = [m for s, m in self.messages if s <= Severity.error]
        return errs[-1] if errs else ""

    def build(self, path: str, config: BuildConfig) -> bool:
        c = _BuildConfig(*[int(getattr(config, f[0])) for f in _BuildConfig._fields_])
        return bool(self._L.w2x_build(self._h, os.fsencode(path), C.byref(c)))

    def load(self, path: str, config: RenderConfig) -> bool:
        c = _RenderConfig(config.deviceId, int(config.precision), config.batchSize, config.channels, config.height,
                          config.width, config.scaling, float(config.overlap[0]), float(config.overlap[1]),
                          int(config.tta), int(config.ttaBugCompat))
        ok = bool(self._L.w2x_load(self._h, os.fsencode(path), C.byref(c)))
        self._scaling = config.scaling if ok else 0
        self._batch, self._tile = (config.batchSize, config.height) if ok else (0, 0)
        return ok

    def render(self, src: np.ndarray, dst: np.ndarray | None = None):
        """render(src, dst) -> bool like the reference; render(src) -> dst array or raises."""
        bps = src.dtype.itemsize                  # uint8 frames, or uint16 ones (extension: w2x_render16)
        if src.dtype not in (np.uint8, np.uint16) or src.ndim != 3 or src.shape[2] != 3 or src.strides[2] != bps or src.strides[1] != 3 * bps:
            raise ValueError("src must be a uint8 (or uint16) [rows, cols, 3] BGR array with packed pixels")
        ret_array = dst is None
        if dst is None:
            s = getattr(self, "_scaling", 0)
            dst = np.empty((src.shape[0] * s, src.shape[1] * s, 3), src.dtype)
        s = getattr(self, "_scaling", 0)
        if s and (dst.dtype != src.dtype or dst.shape != (src.shape[0] * s, src.shape[1] * s, 3) or dst.strides[2] != bps or dst.strides[1] != 3 * bps):
            # the C ABI only sees pointers and steps, so the cv::Mat-style size check lives here
            self._on_msg(int(Severity.error), f"[render@0] Output image has invalid size: expected {src.shape[1] * s}x{src.shape[0] * s}.".encode(), None)
            return False
        fn = self._L.w2x_render if bps == 1 else self._L.w2x_render16
        ok = bool(fn(self._h, src.ctypes.data, src.shape[0], src.shape[1], src.strides[0], dst.ctypes.data, dst.strides[0]))
        if ret_array:
            if not ok:
                raise W2xError(self.last_error() or "render failed")
            return dst
        return ok

    def render_strip(self, src: np.ndarray, dst: np.ndarray, part: int, parts: int) -> bool:
        """One device's share of a frame split into tile-column strips (w2x_render_strip): writes only its columns of dst."""
        s = getattr(self, "_scaling", 0)
        if src.dtype != np.uint8 or src.ndim != 3 or src.shape[2] != 3 or src.strides[2] != 1 or src.strides[1] != 3:
            raise ValueError("src must be a uint8 [rows, cols, 3] BGR array with packed pixels")
        if dst.dtype != np.uint8 or dst.shape != (src.shape[0] * s, src.shape[1] * s, 3) or dst.strides[2] != 1 or dst.strides[1] != 3:
            raise ValueError("dst must be a packed uint8 array of the scaled size")
        return bool(self._L.w2x_render_strip(self._h, src.ctypes.data, src.shape[0], src.shape[1], src.strides[0],
                                             dst.ctypes.data, dst.strides[0], int(part), int(parts)))

    # ---- one frame over several PROCESSES (one engine each): w2x_shard_compute / w2x_shard_slab / w2x_shard_finish + the IPC helpers below
    def shard_compute(self, src: np.ndarray, part: int, parts: int) -> bool:
        if src.dtype != np.uint8 or src.ndim != 3 or src.shape[2] != 3 or src.strides[2] != 1 or src.strides[1] != 3:
            raise ValueError("src must be a uint8 [rows, cols, 3] BGR array with packed pixels")
        return bool(self._L.w2x_shard_compute(self._h, src.ctypes.data, src.shape[0], src.shape[1], src.strides[0], int(part), int(parts)))

    def shard_slab_handle(self):
        """(device pointer, 64-byte IPC handle) of this engine's tile slab (changes only when a larger frame makes the slab grow)"""
        ptr = self._L.w2x_shard_slab(self._h, None)
        buf = (C.c_uint8 * 64)()
        if not ptr or not self._L.w2x_ipc_export(ptr, buf):
            raise W2xError("could not export the tile slab (hipIpcGetMemHandle)")
        return int(ptr), bytes(buf)

    def shard_finish(self, dst: np.ndarray, part: int, parts: int, slabs, devices=None) -> bool:
        """slabs[q]: device pointer (int) of part q's slab in THIS process (w2x ipc_open of its handle; 0 for parts that are not needed)"""
        if dst.dtype != np.uint8 or dst.ndim != 3 or dst.shape[2] != 3 or dst.strides[2] != 1 or dst.strides[1] != 3:
            raise ValueError("dst must be a packed uint8 [rows, cols, 3] array of the scaled size")
        arr = (C.c_void_p * parts)(*[C.c_void_p(int(p) or None) for p in slabs])
        dev = (C.c_int * parts)(*[int(d) for d in devices]) if devices is not None else None
        return bool(self._L.w2x_shard_finish(self._h, dst.ctypes.data, dst.shape[0], dst.shape[1], dst.strides[0], int(part), int(parts), arr, dev))

    def alloc_host(self, shape) -> np.ndarray:
        """A uint8 array over page-locked memory owned by the engine (w2x_alloc_host): frame buffers whose PCIe copies
        render_sequence() can overlap with the kernels.  Valid until free_host(arr) / close()."""
        n = int(np.prod(shape))
        ptr = self._L.w2x_alloc_host(self._h, n)
        if not ptr:
            raise W2xError("w2x_alloc_host failed")
        arr = np.frombuffer((C.c_uint8 * n).from_address(ptr), np.uint8).reshape(shape)
        self._host_bufs = getattr(self, "_host_bufs", {})
        self._host_bufs[arr.ctypes.data] = ptr
        return arr

    def free_host(self, arr: np.ndarray) -> None:
        ptr = getattr(self, "_host_bufs", {}).pop(arr.ctypes.data, None)
        if ptr:
            self._L.w2x_free_host(self._h, ptr)

    def render_sequence(self, frames, outs=None, pinned: bool = False):
        """Equally sized frames with upload / compute / download overlapped (w2x_render_sequence).  outs: list of pre-allocated
        arrays (may repeat, e.g. a ring of buffers) or None; pinned=True takes the output buffers it allocates from alloc_host()
        (copies of the results are returned) - pass alloc_host() arrays as frames / outs yourself to avoid that copy."""
        s = getattr(self, "_scaling", 0)
        n = len(frames)
        if n == 0:
            return []
        r, c = frames[0].shape[:2]
        for f in frames:
            if f.dtype != np.uint8 or f.shape != (r, c, 3) or f.strides != (c * 3, 3, 1):
                raise ValueError("frames must be packed uint8 [rows, cols, 3] arrays of one size")
        own = []
        if outs is None:
            if pinned:
                own = [self.alloc_host((r * s, c * s, 3)) for _ in range(min(n, 3))]
                outs = [own[k % len(own)] for k in range(n)]
            else:
                outs = [np.empty((r * s, c * s, 3), np.uint8) for _ in range(n)]
        for o in outs:
            if o.dtype != np.uint8 or o.shape != (r * s, c * s, 3) or o.strides != (c * s * 3, 3, 1):
                raise ValueError("outs must be packed uint8 arrays of the scaled size")
        import ctypes as C
        sp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        dp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        if own:          # a ring of three engine-owned buffers: run the sequence in pieces and copy each result out
            res = []
            try:
                for k0 in range(0, n, len(own)):
                    m = min(len(own), n - k0)
                    if not self._L.w2x_render_sequence(self._h, (C.c_void_p * m)(*[f.ctypes.data for f in frames[k0:k0 + m]]), r, c, c * 3,
                                                       (C.c_void_p * m)(*[o.ctypes.data for o in own[:m]]), c * s * 3, m):
                        raise W2xError(self.last_error() or "render_sequence failed")
                    res += [o.copy() for o in own[:m]]
            finally:
                for o in own:
                    self.free_host(o)
            return res
        if not self._L.w2x_render_sequence(self._h, sp, r, c, c * 3, dp, c * s * 3, n):
            raise W2xError(self.last_error() or "render_sequence failed")
        return outs

    def infer(self, x: np.ndarray) -> np.ndarray:
        """Private trt::Img2Img::infer (img2img_infer.cpp:41-93) as a test hook: [B,3,T,T] f32 -> [B,3,T',T'] f32."""
        x = np.ascontiguousarray(x, np.float32)
        b, t = getattr(self, "_batch", 0), getattr(self, "_tile", 0)
        if not b:
            raise W2xError("infer called before a successful load")
        if x.shape != (b, 3, t, t):        # img2img_infer.cpp:43-68: batch count and tile shape must be the loaded configuration's
            raise ValueError(f"infer expects a [{b}, 3, {t}, {t}] blob, got {list(x.shape)}")
        to = self.output_tile_size
        y = np.empty((b, 3, to, to), np.float32)
        if not self._L.w2x_infer(self._h, x.ctypes.data, y.ctypes.data):
            raise W2xError(self.last_error() or "infer failed")
        return y

    @property
    def output_tile_size(self) -> int:
        return self._L.w2x_output_tile_size(self._h)

    @property
    def pass_tiles(self) -> int:
        return self._L.w2x_pass_tiles(self._h)

    @property
    def plan_flops(self) -> float:
        return self._L.w2x_plan_flops(self._h)

    @property
    def last_render_ms(self) -> float:
        return self._L.w2x_last_render_ms(self._h)

    def bench_resident(self, iters: int) -> float:
        return self._L.w2x_bench_resident(self._h, iters)

    def profile_frame(self) -> dict:
        """HIP-event time per kernel family for one resident frame: {family: (ms, launches, flop)}, plus 'frame_ms'."""
        out = np.zeros(31, np.float64)
        if not self._L.w2x_profile_frame(self._h, out.ctypes.data, 31):
            raise W2xError(self.last_error() or "profile failed")
        names = ["gemm", "attention", "se_scale", "gather", "compose", "mlp"]
        d = {n: (float(out[5 * i]), int(out[5 * i + 1]), float(out[5 * i + 2])) for i, n in enumerate(names)}
        d["frame_ms"] = float(out[30])
        return d

    def op_times(self) -> np.ndarray:
        out = np.zeros(4096, np.float64)
        n = self._L.w2x_op_times(self._h, out.ctypes.data, 4096)
        return out[:n].copy()


def render_sharded(engines, src: np.ndarray, dst: np.ndarray = None) -> np.ndarray:
    """ONE frame over several engines of this process, every tile computed once (w2x_render_sharded): engine k takes the k-th contiguous
    range of the tile order, the seam bands travel device to device, each engine composes and downloads its own cells of `dst`."""
    s = getattr(engines[0], "_scaling", 0)
    if src.dtype != np.uint8 or src.ndim != 3 or src.shape[2] != 3 or src.strides[2] != 1 or src.strides[1] != 3:
        raise ValueError("src must be a uint8 [rows, cols, 3] BGR array with packed pixels")
    if dst is None:
        dst = np.empty((src.shape[0] * s, src.shape[1] * s, 3), np.uint8)
    if dst.dtype != np.uint8 or dst.shape != (src.shape[0] * s, src.shape[1] * s, 3) or dst.strides[2] != 1 or dst.strides[1] != 3:
        raise ValueError("dst must be a packed uint8 array of the scaled size")
    handles = (C.c_void_p * len(engines))(*[e._h for e in engines])
    if not lib().w2x_render_sharded(handles, len(engines), src.ctypes.data, src.shape[0], src.shape[1], src.strides[0], dst.ctypes.data, dst.strides[0]):
        raise W2xError(engines[0].last_error() or "sharded render failed")
    return dst


class debug_switches:
    """Test hook (w2x_debug_set, csrc/switches.h): the reference paths the A/B tests compare the shipped plans and kernels with -
    `with debug_switches(no_fuse_attn=1): eng.build(...)`.  Process-wide; every switch goes back to 0 on exit."""

    def __init__(self, **switches):
        self.switches = switches

    def __enter__(self):
        for k, v in self.switches.items():
            if not lib().w2x_debug_set(k.encode(), int(v)):
                raise W2xError(f"no such switch: {k}")
        return self

    def __exit__(self, *exc):
        for k in self.switches:
            lib().w2x_debug_set(k.encode(), 0)
        return False


def device_pci_bus_id(device: int):
    """PCI bus id of HIP device `device` of this process (W2X_DEVICE_MAP applied), or None: w2x_device_pci_bus_id.  Initialises the HIP runtime."""
    buf = C.create_string_buffer(64)
    return buf.value.decode() if lib().w2x_device_pci_bus_id(int(device), buf, 64) else None


def ipc_open(handle: bytes, device: int) -> int:
    """open another process's 64-byte device-memory handle on logical device `device` -> device pointer (0: failed)"""
    buf = (C.c_uint8 * 64)(*handle)
    return int(lib().w2x_ipc_open(buf, int(device)) or 0)


def ipc_close(ptr: int) -> None:
    if ptr:
        lib().w2x_ipc_close(C.c_void_p(ptr))


def shard_plan(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, overlap, part, parts):
    """Host logic of the every-tile-once split of one frame -> (first_tile, tile_count, halo_first, [(x, y, w, h), ...])."""
    out = np.zeros(16, np.int32)
    lib().w2x_shard_plan(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, float(overlap[0]), float(overlap[1]), int(part), int(parts), out.ctypes.data)
    return int(out[0]), int(out[1]), int(out[2]), [tuple(int(v) for v in out[4 + 4 * r:8 + 4 * r]) for r in range(int(out[3]))]


def strip_plan(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, overlap, part, parts):
    """Host logic of the multi-GPU single-frame split -> (first_tile, tile_count, x0, x1)."""
    out = np.zeros(4, np.int32)
    lib().w2x_strip_plan(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, float(overlap[0]), float(overlap[1]), int(part), int(parts), out.ctypes.data)
    return tuple(int(v) for v in out)


def calculate_tiles(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, overlap):
    """calculateTiles (img2img_render.cpp:7-66) through the C ABI -> (count, in_rects[N,4], out_rects[N,4])."""
    L = lib()
    cap = 1 << 16
    a = np.zeros((cap, 4), np.int32)
    b = np.zeros((cap, 4), np.int32)
    n = L.w2x_calculate_tiles(in_w, in_h, out_w, out_h, tile_in, tile_out, scaling, float(overlap[0]), float(overlap[1]),
                              a.ctypes.data, b.ctypes.data, cap)
    if n < 0:
        raise W2xError("tile capacity exceeded")
    return n, a[:n].copy(), b[:n].copy()


def tile_weights(which, ovx, ovy, size):
    L = lib()
    out = np.empty((size, size), np.float32)
    if not L.w2x_tile_weights(which, ovx, ovy, size, out.ctypes.data):
        raise W2xError("bad arguments")
    return out


def describe_plan(onnx_path, batch, tile, precision=None) -> str:
    L = lib()
    buf = C.create_string_buffer(1 << 20)
    ok = L.w2x_describe_plan_precision(os.fsencode(onnx_path), batch, tile, int(Precision.FP16 if precision is None else precision), buf, len(buf))
    s = buf.value.decode(errors="replace")
    if not ok:
        raise W2xError(s)
    return s


def write_engine_file(onnx_path, batch, tile, out_path) -> bool:
    """Host-only half of build(): lower the graph and write the plan file (no device needed)."""
    return bool(lib().w2x_write_engine_file(os.fsencode(onnx_path), batch, tile, os.fsencode(out_path)))


def validate_engine_file(path) -> tuple[bool, str]:
    """Host-only half of load(): deserialize + consistency checks -> (ok, reason)."""
    buf = C.create_string_buffer(4096)
    ok = lib().w2x_validate_engine_file(os.fsencode(path), buf, len(buf))
    return bool(ok), buf.value.decode(errors="replace")


def sha256_hex(data: bytes) -> str:
    L = lib()
    out = C.create_string_buffer(65)
    L.w2x_sha256_hex(data, len(data), out)
    return out.value.decode()
