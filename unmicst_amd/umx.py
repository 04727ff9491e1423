"""ctypes binding of libumx.so (include/umx.h) -- the thin shim between Python and the HIP engine.

There is deliberately no fallback: if the shared library is missing or no gfx950 device is present the calls
raise, they never route to a CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import numpy as np

from . import build as _build
from .model import HParams

UMX_OK = 0
PREC_DEFAULT, PREC_F32, PREC_F16X3, PREC_F16X3_F6 = 0, 1, 2, 3      # enum umx_precision
PRECISIONS = {"default": PREC_DEFAULT, "f32": PREC_F32, "f16x3": PREC_F16X3, "f16f6": PREC_F16X3_F6}
ERR_INVALID, ERR_HIP = 1, 4   # UMX_ERR_INVALID, UMX_ERR_HIP
ERR_RANGE = 6   # UMX_ERR_RANGE
MODE_ACCUMULATE, MODE_REPLACE = 0, 1
STITCH_FP16_COMPAT, STITCH_FP32 = 0, 1

# every symbol include/umx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "umx_device_count", "umx_device_mem_info", "umx_create", "umx_create_opts", "umx_precision_of", "umx_destroy", "umx_last_error", "umx_set_stream", "umx_synchronize",
    "umx_forward_tiles", "umx_forward_tiles_dev", "umx_tile_grid", "umx_infer_image", "umx_infer_image_dev",
    "umx_infer_image_raw", "umx_infer_image_raw_range", "umx_plane_range", "umx_infer_image_raw_scaled", "umx_infer_image_raw_outlier", "umx_infer_image_raw_submit", "umx_infer_image_wait", "umx_tiff_lzw_decode",
    "umx_tiff_packbits_decode", "umx_shard_unique_id", "umx_shard_init", "umx_shard_init_transport", "umx_shard_fini", "umx_shard_plan",
    "umx_infer_image_sharded_dev", "umx_infer_image_sharded_raw", "umx_infer_image_sharded_raw_submit",
    "umx_band_tiles_dev", "umx_stitch_dev", "umx_profile_enable", "umx_profile_read", "umx_prof_entry_size", "umx_test_double_to_half", "umx_test_double_to_half_dev",
    "umx_describe", "umx_describe_graph", "umx_plan_check", "umx_test_mx_pack_e2m3", "umx_version",
]


class UmxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libumx error %d: %s" % (code, msg))
        self.code = code


class _HP(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("graph", "imSize", "nChannels", "nClasses", "nOut0", "nLayers", "ks", "nExtraConvs", "featMapsFact")]


class _Options(ctypes.Structure):
    _fields_ = [("device_ordinal", ctypes.c_int32), ("max_batch", ctypes.c_int32), ("precision", ctypes.c_int32),
                ("act_shift", ctypes.c_int32), ("lanes", ctypes.c_int32), ("reserved", ctypes.c_int32 * 11)]


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("kernel", ctypes.c_char * 64), ("launches", ctypes.c_int64),
                ("total_ms", ctypes.c_double), ("flops", ctypes.c_double), ("bytes", ctypes.c_double),
                ("exec_flops", ctypes.c_double), ("launches_seen", ctypes.c_int64), ("xcd_order", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


_SEND_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p)
_GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
_GROUP_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)


class ShardTransport(ctypes.Structure):
    """umx_shard_transport (include/umx.h): the inter-rank operations of umx_infer_image_sharded_dev as a table of callbacks."""
    _fields_ = [("user", ctypes.c_void_p), ("send", _SEND_FN), ("recv", _SEND_FN), ("all_gather", _GATHER_FN),
                ("group_start", _GROUP_FN), ("group_end", _GROUP_FN)]


_lib = None
hip_runtime = None      # path of the libamdhip64 libumx was bound to (set by load())


def _torch_hip_runtime() -> Optional[str]:
    """Path of the libamdhip64 PyTorch bundles, WITHOUT importing torch (find_spec only locates the package)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    return cand if os.path.exists(cand) else None


def _bind_hip_runtime() -> str:
    """Make exactly one HIP runtime (libamdhip64) global in this process before libumx is dlopen'ed.

    libumx.so is linked without a HIP runtime dependency.  PyTorch wheels bundle their own libamdhip64 +
    libhsa-runtime64; a second copy (e.g. /opt/rocm's) initialised in the same process leaves one of the two
    without a GPU.  Policy (UMX_HIP_RUNTIME = auto | torch | system, default auto):
      auto    PyTorch's copy whenever a PyTorch with a bundled runtime is INSTALLED (located with find_spec and dlopen'ed
              directly: torch itself is not imported, so the choice does not depend on import order and costs nothing) --
              a later `import torch` in the process (sharding.py, the trainer's helpers, the oracles) then finds the very
              runtime libumx already uses; the system ROCm runtime otherwise.
      system  the system ROCm runtime.  The per-file command-line tools ask for it (driver.main: they never import torch, and
              the system runtime initialises 0.1 s faster); mixing it with a later `import torch` is refused by
              require_torch_runtime().
      torch   import torch first (as rounds 1-4 did under the tests), then bind its copy.
    """
    import sys
    mode = os.environ.get("UMX_HIP_RUNTIME", "auto")
    if mode not in ("auto", "torch", "system"):
        raise ValueError("UMX_HIP_RUNTIME must be auto, torch or system (got %r)" % mode)
    if mode == "torch":
        import torch  # noqa: F401  (loads its bundled libamdhip64.so)
    if mode in ("auto", "torch"):
        cand = _torch_hip_runtime()
        if cand is not None:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            return cand
        if "torch" in sys.modules and mode == "torch":
            raise OSError("UMX_HIP_RUNTIME=torch: this PyTorch bundles no libamdhip64")
    for cand in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"), "libamdhip64.so.7",
                 "libamdhip64.so"):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            return cand
        except OSError:
            continue
    raise OSError("no HIP runtime (libamdhip64) found; libumx has no CPU fallback")


def require_torch_runtime(who: str) -> None:
    """Called by the torch-facing helpers (unmicst_amd/sharding.py, bench.py) before they hand torch tensors to libumx:
    refuses to run when libumx is bound to another HIP runtime than the one PyTorch brings -- two runtimes in one process and
    one of them loses the GPU (the symptom is hipErrorNoDevice or a hang far from the cause)."""
    load()
    want = _torch_hip_runtime()
    if want is not None and hip_runtime is not None and os.path.realpath(want) != os.path.realpath(hip_runtime):
        raise RuntimeError("%s needs libumx and PyTorch on ONE HIP runtime, but libumx is bound to %s and PyTorch brings %s "
                           "(UMX_HIP_RUNTIME=%s): leave UMX_HIP_RUNTIME unset (auto) in processes that import torch"
                           % (who, hip_runtime, want, os.environ.get("UMX_HIP_RUNTIME", "auto")))


def load(path: Optional[str] = None):
    """dlopen libumx.so (built in-tree by unmicst_amd.build); raises if it is missing."""
    global _lib, hip_runtime
    if _lib is not None:
        return _lib
    path = path or os.environ.get("UMX_LIB") or _build.lib_path()   # UMX_LIB: an alternative build (kernel experiments)
    if not os.path.exists(path):
        raise FileNotFoundError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(libumx has no CPU fallback)" % path)
    hip_runtime = _bind_hip_runtime()
    L = ctypes.CDLL(path)
    c_int, c_void_p, c_double = ctypes.c_int, ctypes.c_void_p, ctypes.c_double
    ip = ctypes.POINTER(c_int)
    L.umx_device_count.restype = c_int
    L.umx_device_mem_info.restype = c_int
    L.umx_device_mem_info.argtypes = [c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    L.umx_create.restype = c_int
    L.umx_create.argtypes = [ctypes.POINTER(_HP), c_void_p, ctypes.c_size_t, c_int, c_int, ctypes.POINTER(c_void_p)]
    L.umx_create_opts.restype = c_int
    L.umx_create_opts.argtypes = [ctypes.POINTER(_HP), c_void_p, ctypes.c_size_t, ctypes.POINTER(_Options),
                                  ctypes.POINTER(c_void_p)]
    L.umx_precision_of.restype = c_int
    L.umx_precision_of.argtypes = [c_void_p]
    L.umx_destroy.restype = None
    L.umx_destroy.argtypes = [c_void_p]
    L.umx_last_error.restype = ctypes.c_char_p
    L.umx_last_error.argtypes = [c_void_p]
    L.umx_set_stream.argtypes = [c_void_p, c_void_p]
    L.umx_synchronize.argtypes = [c_void_p]
    L.umx_forward_tiles.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
    L.umx_forward_tiles_dev.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
    L.umx_tile_grid.argtypes = [c_void_p, c_int, c_int, ip, ip, ip, ip]
    L.umx_infer_image.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_int, c_int, c_void_p]
    L.umx_infer_image_dev.argtypes = L.umx_infer_image.argtypes
    L.umx_infer_image_raw_outlier.restype = c_int
    L.umx_infer_image_raw_outlier.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_double, c_double,
                                              c_int, c_void_p]
    L.umx_infer_image_raw_scaled.restype = c_int
    L.umx_infer_image_raw_scaled.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_double, c_double,
                                             c_int, c_void_p]
    L.umx_infer_image_raw_submit.restype = c_int
    L.umx_infer_image_raw_submit.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_double,
                                             c_int, c_void_p]
    L.umx_infer_image_wait.restype = c_int
    L.umx_infer_image_wait.argtypes = [c_void_p, c_int]
    L.umx_shard_unique_id.restype = c_int
    L.umx_shard_unique_id.argtypes = [c_void_p]
    L.umx_shard_init.restype = c_int
    L.umx_shard_init.argtypes = [c_void_p, c_void_p, c_int, c_int]
    L.umx_shard_fini.restype = c_int
    L.umx_shard_fini.argtypes = [c_void_p]
    L.umx_shard_init_transport.restype = c_int
    L.umx_shard_init_transport.argtypes = [c_void_p, ctypes.POINTER(ShardTransport), c_int, c_int]
    L.umx_shard_plan.restype = c_int
    L.umx_shard_plan.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 9
    L.umx_infer_image_sharded_dev.restype = c_int
    L.umx_infer_image_sharded_dev.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_int,
                                              c_int, c_int, c_void_p]
    L.umx_infer_image_sharded_raw_submit.restype = c_int
    L.umx_infer_image_sharded_raw_submit.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                     c_double, c_double, c_int, c_int, c_void_p, c_void_p]
    L.umx_infer_image_sharded_raw.restype = c_int
    L.umx_infer_image_sharded_raw.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                              c_double, c_double, c_int, c_int, c_void_p, c_void_p]
    for fn in (L.umx_tiff_lzw_decode, L.umx_tiff_packbits_decode):
        fn.restype = ctypes.c_longlong
        fn.argtypes = [c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t]
    L.umx_infer_image_raw.restype = c_int
    L.umx_infer_image_raw.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_int,
                                      c_void_p]
    L.umx_plane_range.restype = c_int
    L.umx_plane_range.argtypes = [c_void_p, c_int, ctypes.c_size_t, c_void_p]
    L.umx_infer_image_raw_range.restype = c_int
    L.umx_infer_image_raw_range.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_double, c_double, c_int,
                                            c_void_p]
    L.umx_band_tiles_dev.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_double,
                                     c_int, c_int, c_void_p]
    L.umx_stitch_dev.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.umx_profile_enable.argtypes = [c_void_p, c_int]
    L.umx_profile_read.argtypes = [c_void_p, ctypes.POINTER(ProfEntry), c_int, ip]
    L.umx_test_double_to_half.restype = None
    L.umx_test_double_to_half.argtypes = [c_void_p, c_void_p, ctypes.c_size_t]
    L.umx_describe.argtypes = [ctypes.POINTER(_HP), ip, ctypes.POINTER(c_double), ctypes.POINTER(c_double)]
    L.umx_describe_graph.argtypes = [ctypes.POINTER(_HP), ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.umx_describe_graph.restype = c_int
    L.umx_version.restype = ctypes.c_char_p
    L.umx_prof_entry_size.restype = c_int
    if L.umx_prof_entry_size() != ctypes.sizeof(ProfEntry):
        raise RuntimeError("libumx (%s) lays umx_prof_entry out in %d bytes, this binding in %d: mixed builds" % (
            L.umx_version().decode(), L.umx_prof_entry_size(), ctypes.sizeof(ProfEntry)))
    for name in ("umx_set_stream", "umx_synchronize", "umx_forward_tiles", "umx_forward_tiles_dev", "umx_tile_grid",
                 "umx_infer_image", "umx_infer_image_dev", "umx_band_tiles_dev", "umx_stitch_dev",
                 "umx_profile_enable", "umx_profile_read", "umx_describe"):
        getattr(L, name).restype = c_int
    _lib = L
    return L


def plane_range(raw: np.ndarray):
    """(min, max) of a contiguous uint8 / uint16 array by umx_plane_range (one pass, host threads; no GPU involved)."""
    a = np.ascontiguousarray(raw)
    if a.dtype not in (np.uint8, np.uint16) or a.size == 0:
        raise TypeError("plane_range takes non-empty uint8 / uint16 arrays")
    a = a.astype(a.dtype.newbyteorder("="), copy=False)
    out = np.zeros(2, np.uint32)
    rc = load().umx_plane_range(a.ctypes.data, a.dtype.itemsize * 8, a.size, out.ctypes.data)
    if rc != 0:
        raise UmxError(rc, "umx_plane_range")
    return int(out[0]), int(out[1])


def _hp_struct(hp: HParams) -> _HP:
    return _HP(hp.graph, hp.imSize, hp.nChannels, hp.nClasses, hp.nOut0, hp.nLayers, hp.ks, hp.nExtraConvs,
               hp.featMapsFact)


def device_count() -> int:
    return int(load().umx_device_count())


def device_mem_info(device: int):
    """(free_bytes, total_bytes) of one device -- the quantity toolbox/GPUselect.py reads through NVML."""
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    rc = load().umx_device_mem_info(int(device), ctypes.byref(f), ctypes.byref(t))
    if rc:
        raise UmxError(rc, load().umx_last_error(None).decode())
    return f.value, t.value


def pick_device_most_free_memory() -> int:
    """HIP analogue of GPUselect.pick_gpu_lowest_memory (reference toolbox/GPUselect.py:4-22)."""
    n = device_count()
    if n < 1:
        raise UmxError(3, "no HIP device available (libumx has no CPU fallback)")
    return max(range(n), key=lambda d: device_mem_info(d)[0])


def describe(hp: HParams) -> dict:
    n, f, e = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    h = _hp_struct(hp)
    rc = load().umx_describe(ctypes.byref(h), ctypes.byref(n), ctypes.byref(f), ctypes.byref(e))
    if rc:
        raise UmxError(rc, load().umx_last_error(None).decode())
    return {"launches": n.value, "flops_per_tile": f.value, "executed_flops_per_tile": e.value}


def describe_graph(hp: HParams) -> dict:
    """The library's wiring of a model (buffers, launch list, constants): see umx_describe_graph in include/umx.h.  Host only."""
    import json
    h = _hp_struct(hp)
    need = ctypes.c_size_t()
    rc = load().umx_describe_graph(ctypes.byref(h), None, 0, ctypes.byref(need))
    if rc:
        raise UmxError(rc, load().umx_last_error(None).decode())
    buf = ctypes.create_string_buffer(need.value)
    rc = load().umx_describe_graph(ctypes.byref(h), buf, need.value, ctypes.byref(need))
    if rc:
        raise UmxError(rc, load().umx_last_error(None).decode())
    return json.loads(buf.value.decode())


def plan_check(hp: HParams) -> str:
    """umx_plan_check: "" if the split-precision planner takes every layer of this model, else the first refused layer and why.  Host only."""
    h = _hp_struct(hp)
    L = load()
    L.umx_plan_check.restype = ctypes.c_int
    L.umx_plan_check.argtypes = [ctypes.c_void_p]
    rc = L.umx_plan_check(ctypes.byref(h))
    return "" if rc == 0 else L.umx_last_error(None).decode()


def auto_batch(hp: HParams, arena_bytes: float = 12 * 2 ** 30) -> int:
    """Tiles per launch group for a model: enough pixels per launch to fill the chip at the deep, small layers (a 64 x 64-pixel
    tile is 4 x 4 pixels at the solo model's bottom: 256 tiles are 160 workgroups on 512 slots) -- 2^24 pixels per group, i.e.
    256 / 1024 / 4096 tiles of 256 / 128 / 64 pixels -- capped so that the activation arena stays within `arena_bytes`.
    Measured on MI355X (profiles/r03/batch_sweep.txt): solo 64-px tiles +12 %, duo 128-px tiles +22 % over groups of 256."""
    g = describe_graph(hp)
    per_tile = sum(b["size"] ** 2 * ((b["channels"] + 7) // 8 * 8) * 4 for b in g["buffers"])
    by_pixels = max(int(hp.batchSize), min(4096, (1 << 24) // (hp.imSize * hp.imSize)))
    cap = max(1, int(arena_bytes // max(per_tile, 1)))
    cap = 1 << (cap.bit_length() - 1)                      # power of two below the cap
    return max(1, min(by_pixels, max(cap, int(hp.batchSize))))


def double_to_half(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(x.shape, np.uint16)
    load().umx_test_double_to_half(x.ctypes.data, out.ctypes.data, x.size)
    return out.view(np.float16)


def double_to_half_dev(x: np.ndarray) -> np.ndarray:
    """umx_test_double_to_half_dev: the DEVICE routine of the stitch kernel (needs a GPU)."""
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(x.shape, np.uint16)
    L = load()
    L.umx_test_double_to_half_dev.restype = ctypes.c_int
    L.umx_test_double_to_half_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rc = L.umx_test_double_to_half_dev(x.ctypes.data, out.ctypes.data, x.size)
    if rc:
        raise UmxError(rc, L.umx_last_error(None).decode())
    return out.view(np.float16)


def tiff_decode(kind: str, buf: bytes, nbytes: int) -> bytes:
    """Decode one TIFF strip / tile: kind "lzw" (compression 5) or "packbits" (32773) -> exactly ``nbytes`` bytes
    (zero-filled if the stream is short)."""
    L = load()
    out = ctypes.create_string_buffer(nbytes)
    fn = {"lzw": L.umx_tiff_lzw_decode, "packbits": L.umx_tiff_packbits_decode}[kind]
    n = fn(buf, len(buf), out, nbytes)
    if n < 0:
        raise ValueError("malformed %s stream in TIFF strip" % kind)
    return out.raw


def shard_plan(hp: HParams, H: int, W: int, rank: int, world: int, nslabs: int = 2, slab: int = 0) -> dict:
    """umx_shard_plan: band / slab geometry of one rank (needs no device)."""
    L = load()
    v = [ctypes.c_int() for _ in range(9)]
    h = _hp_struct(hp)
    rc = L.umx_shard_plan(ctypes.byref(h), H, W, int(rank), int(world), int(nslabs), int(slab), *[ctypes.byref(x) for x in v])
    if rc:
        raise UmxError(rc, L.umx_last_error(None).decode())
    names = ("patch_row0", "patch_row1", "need_row0", "need_row1", "own_row0", "own_row1", "slab_row0", "slab_row1", "nslabs")
    return {n: x.value for n, x in zip(names, v)}


class Engine:
    """One umx_ctx: a model resident on one MI355X."""

    def __init__(self, hp: HParams, blob: np.ndarray, device: int = 0, max_batch: int = 32,
                 precision="default", act_shift: int = -1, lanes: int = 0):
        """precision: "default" (f16x3 unless UMX_PRECISION=f32), "f32" (exact fp32 MFMA) or "f16x3" (three binary16
        MFMA products per fp32 product, fp32 accumulation) -- both hold the 1e-4 tolerance.
        lanes: 0 = library default, 1 or 2 activation-buffer sets / streams the tile batches alternate between."""
        self._L = load()
        self.hp = hp
        self._ctx = ctypes.c_void_p()
        blob = np.ascontiguousarray(blob, dtype="<f4")
        h = _hp_struct(hp)
        o = _Options(int(device), int(max_batch), PRECISIONS.get(precision, precision), int(act_shift), int(lanes))
        rc = self._L.umx_create_opts(ctypes.byref(h), blob.ctypes.data, blob.size, ctypes.byref(o),
                                     ctypes.byref(self._ctx))
        if rc:
            raise UmxError(rc, self._L.umx_last_error(None).decode())
        self.device = device
        self.max_batch = max_batch
        self.precision = {PREC_F32: "f32", PREC_F16X3: "f16x3", PREC_F16X3_F6: "f16f6"}[self._L.umx_precision_of(self._ctx)]
        self.note = ""
        if precision == "default" and self.precision == "f32":   # the planner of the fast kernels refused the model: say so
            import warnings
            self.note = self._L.umx_last_error(self._ctx).decode()
            if self.note.startswith("note:"):   # (not when UMX_PRECISION=f32 asked for it)
                warnings.warn(self.note, RuntimeWarning, stacklevel=2)

    # -- lifetime
    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.umx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int) -> None:
        if rc:
            raise UmxError(rc, self._L.umx_last_error(self._ctx).decode())

    # -- plumbing
    # -- multi-GPU inside the library (RCCL over xGMI): see include/umx.h
    @staticmethod
    def shard_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        rc = load().umx_shard_unique_id(buf)
        if rc:
            raise UmxError(rc, load().umx_last_error(None).decode())
        return buf.raw

    def shard_init(self, unique_id: bytes, rank: int, world: int) -> None:
        self._check(self._L.umx_shard_init(self._ctx, ctypes.create_string_buffer(unique_id, 128), int(rank), int(world)))

    def shard_fini(self) -> None:
        """umx_shard_fini: drop the communicator / transport (after a failure: ncclCommAbort instead of ncclCommDestroy)."""
        self._check(self._L.umx_shard_fini(self._ctx))

    def shard_init_transport(self, send, recv, all_gather, rank: int, world: int, group_start=None, group_end=None) -> None:
        """umx_shard_init_transport with Python callables: send(dev_ptr, nbytes, peer, stream), recv(dev_ptr, nbytes, peer, stream),
        all_gather(send_ptr, recv_ptr, nbytes_per_rank, stream), group_start() / group_end(); an exception = failure."""
        def guard(fn):
            def call(user, *a):
                try:
                    fn(*a)
                    return 0
                except Exception:   # noqa: BLE001  (nothing may propagate through the C frames)
                    import traceback
                    traceback.print_exc()
                    return 1
            return call
        t = ShardTransport(None, _SEND_FN(guard(send)), _SEND_FN(guard(recv)), _GATHER_FN(guard(all_gather)),
                           _GROUP_FN(guard(group_start)) if group_start else _GROUP_FN(), _GROUP_FN(guard(group_end)) if group_end else _GROUP_FN())
        self._transport = t   # (the callbacks must outlive the context's use of them)
        self._check(self._L.umx_shard_init_transport(self._ctx, ctypes.byref(t), int(rank), int(world)))

    def shard_plan(self, H: int, W: int, rank: int, world: int, nslabs: int = 2, slab: int = 0) -> dict:
        return shard_plan(self.hp, H, W, rank, world, nslabs, slab)

    def infer_image_sharded_dev(self, band_ptr: int, C: int, H: int, W: int, band_row0: int, band_rows: int, mean: float,
                                std: float, mode: int, stitch: int, nslabs: int, out_full_ptr: int) -> None:
        self._check(self._L.umx_infer_image_sharded_dev(self._ctx, ctypes.c_void_p(band_ptr), C, H, W, int(band_row0),
                                                        int(band_rows), float(mean), float(std), int(mode), int(stitch),
                                                        int(nslabs), ctypes.c_void_p(out_full_ptr)))

    def infer_image_sharded_raw_submit(self, slot: int, band_ptr: int, bits: int, C: int, H: int, W: int, band_row0: int,
                                       band_rows: int, value_range, mean: float, std: float, mode: int, nslabs: int,
                                       own_out_ptr: int = 0, out_full_ptr: int = 0) -> None:
        """umx_infer_image_sharded_raw_submit: this rank's raw rows (host pointer) in, uint8 planes out -- the rank's own rows to
        ``own_out_ptr`` (host, [K, own rows, W]) and the gathered stack to ``out_full_ptr`` (device, [K, H, W]); 0 = none / the
        library's buffer.  ``value_range``: None, or per plane (min, max) of the WHOLE plane (the drivers' rescale).
        ``infer_image_wait(slot)`` completes the call."""
        rng = None
        if value_range is not None:
            rng = (ctypes.c_uint32 * (2 * C))(*[int(v) for pair in value_range for v in pair])
        self._check(self._L.umx_infer_image_sharded_raw_submit(
            self._ctx, int(slot), ctypes.c_void_p(band_ptr), int(bits), C, H, W, int(band_row0), int(band_rows), rng, float(mean),
            float(std), int(mode), int(nslabs), ctypes.c_void_p(own_out_ptr or None), ctypes.c_void_p(out_full_ptr or None)))

    def infer_image_sharded_raw(self, band: np.ndarray, H: int, W: int, band_row0: int, value_range, mean: float, std: float,
                                mode: int = MODE_ACCUMULATE, nslabs: int = 2, own_rows: int = 0, out_full_ptr: int = 0):
        """Synchronous form on a numpy band [C, rows, W] (or [rows, W]) of uint8 / uint16: returns this rank's own rows as uint8
        [K, own_rows, W] (``own_rows`` from shard_plan: own_row1 - own_row0)."""
        b = np.ascontiguousarray(band)
        if b.ndim == 2:
            b = b[None]
        if b.dtype not in (np.uint8, np.uint16):
            raise TypeError("raw planes must be uint8 or uint16")
        out = np.empty((self.hp.nClasses, int(own_rows), W), dtype=np.uint8)
        self.infer_image_sharded_raw_submit(0, b.ctypes.data, 8 * b.dtype.itemsize, b.shape[0], H, W, band_row0, b.shape[1], value_range,
                                            mean, std, mode, nslabs, out.ctypes.data if out.size else 0, out_full_ptr)
        self.infer_image_wait(0)
        return out

    def set_stream(self, hip_stream) -> None:
        """Run the engine's launches on the caller's HIP stream (a non-zero hipStream_t handle, e.g.
        ``torch.cuda.Stream().cuda_stream``); ``None`` restores the engine's own stream.  The legacy default stream
        (handle 0 -- what ``torch.cuda.current_stream()`` is unless a ``torch.cuda.stream`` context is active) is
        refused: the C ABI reads NULL as "own stream", and that stream is non-blocking, so engine work would NOT be
        ordered against the caller's default-stream work."""
        if hip_stream is None:
            hip_stream = 0
        elif int(hip_stream) == 0:
            raise ValueError("set_stream(0): the legacy default stream cannot be shared with the engine; run the caller's "
                             "work under a dedicated torch.cuda.Stream and pass its cuda_stream handle (None = own stream)")
        self._check(self._L.umx_set_stream(self._ctx, ctypes.c_void_p(int(hip_stream))))

    def synchronize(self) -> None:
        self._check(self._L.umx_synchronize(self._ctx))

    def tile_grid(self, H: int, W: int):
        a, b, c, d = (ctypes.c_int() for _ in range(4))
        self._check(self._L.umx_tile_grid(self._ctx, H, W, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c),
                                          ctypes.byref(d)))
        return a.value, b.value, c.value, d.value

    # -- host-pointer API
    def forward_tiles(self, tiles: np.ndarray) -> np.ndarray:
        """== Session.run(UNet2D.nn, {tfData: tiles, tfTraining: 0}) (reference UnMicst1-5.py:704)."""
        hp = self.hp
        tiles = np.ascontiguousarray(tiles, dtype=np.float32)
        if tiles.ndim != 4 or tiles.shape[1:] != (hp.imSize, hp.imSize, hp.nChannels):
            raise ValueError("tiles must be [n,%d,%d,%d]" % (hp.imSize, hp.imSize, hp.nChannels))
        out = np.empty((tiles.shape[0], hp.imSize, hp.imSize, hp.nClasses), np.float32)
        self._check(self._L.umx_forward_tiles(self._ctx, tiles.ctypes.data, tiles.shape[0], out.ctypes.data))
        return out

    def infer_image(self, image: np.ndarray, mean: float, std: float, mode: int = MODE_ACCUMULATE,
                    stitch: int = STITCH_FP16_COMPAT) -> np.ndarray:
        """All-class whole-image inference: image float64 (H,W) or (C,H,W) -> [K,H,W] float16 / float32."""
        image = np.ascontiguousarray(image, dtype=np.float64)
        if image.ndim == 2:
            image = image[None]
        if image.ndim != 3:
            raise ValueError("image must be (H,W) or (C,H,W)")
        C, H, W = image.shape
        out = np.empty((self.hp.nClasses, H, W), np.float32 if stitch == STITCH_FP32 else np.float16)
        self._check(self._L.umx_infer_image(self._ctx, image.ctypes.data, C, H, W, float(mean), float(std), int(mode),
                                            int(stitch), out.ctypes.data))
        return out

    def infer_image_ptr(self, image_ptr: int, C: int, H: int, W: int, mean: float, std: float, out_ptr: int,
                        mode: int = MODE_ACCUMULATE, stitch: int = STITCH_FP16_COMPAT) -> None:
        """umx_infer_image on raw HOST addresses (e.g. pinned buffers: the slab-wise uploads / downloads are then true DMA
        under the tile kernels): float64 [C,H,W] in, [K,H,W] float16 / float32 out."""
        self._check(self._L.umx_infer_image(self._ctx, ctypes.c_void_p(image_ptr), C, H, W, float(mean), float(std),
                                            int(mode), int(stitch), ctypes.c_void_p(out_ptr)))

    def infer_image_raw_range_ptr(self, raw_ptr: int, bits: int, C: int, H: int, W: int, value_range, mean: float, std: float,
                                  out_ptr: int, mode: int = MODE_ACCUMULATE) -> None:
        """umx_infer_image_raw_range on raw HOST addresses (rescale implied; value_range: per plane (min, max))."""
        rng = np.ascontiguousarray(np.asarray(value_range, dtype=np.uint32).reshape(C, 2))
        self._check(self._L.umx_infer_image_raw_range(self._ctx, ctypes.c_void_p(raw_ptr), int(bits), C, H, W, rng.ctypes.data,
                                                      float(mean), float(std), int(mode), ctypes.c_void_p(out_ptr)))

    def infer_image_raw_ptr(self, raw_ptr: int, bits: int, C: int, H: int, W: int, rescale: bool, mean: float, std: float,
                            out_ptr: int, mode: int = MODE_ACCUMULATE) -> None:
        """umx_infer_image_raw on raw HOST addresses: uint8 / uint16 [C,H,W] in, uint8 [K,H,W] out."""
        self._check(self._L.umx_infer_image_raw(self._ctx, ctypes.c_void_p(raw_ptr), int(bits), C, H, W, int(bool(rescale)),
                                                float(mean), float(std), int(mode), ctypes.c_void_p(out_ptr)))

    def infer_image_raw_submit(self, slot: int, raw_ptr: int, bits: int, C: int, H: int, W: int, rescale: bool, mean: float,
                               std: float, out_ptr: int, mode: int = MODE_ACCUMULATE) -> None:
        """Enqueue one slide on `slot` (0 / 1) and return; pair with infer_image_wait(slot).  HOST addresses."""
        self._check(self._L.umx_infer_image_raw_submit(self._ctx, int(slot), ctypes.c_void_p(raw_ptr), int(bits), C, H, W,
                                                       int(bool(rescale)), float(mean), float(std), int(mode),
                                                       ctypes.c_void_p(out_ptr)))

    def infer_image_wait(self, slot: int) -> None:
        self._check(self._L.umx_infer_image_wait(self._ctx, int(slot)))

    def infer_image_raw(self, raw: np.ndarray, rescale: bool, mean: float, std: float,
                        mode: int = MODE_ACCUMULATE, value_range=None) -> np.ndarray:
        """Driver fast path at scalingFactor 1: raw uint8/uint16 (H,W) or (C,H,W) -> uint8 [K,H,W] (see include/umx.h).
        value_range (with rescale): per plane (min, max) of the raw samples, as the reader found them (umx_infer_image_raw_range:
        the upload then overlaps the tile kernels in this synchronous call too)."""
        raw = np.ascontiguousarray(raw)
        if raw.dtype not in (np.uint8, np.uint16):
            raise TypeError("raw planes must be uint8 or uint16")
        if raw.ndim == 2:
            raw = raw[None]
        if raw.ndim != 3:
            raise ValueError("raw must be (H,W) or (C,H,W)")
        C, H, W = raw.shape
        raw = raw.astype(raw.dtype.newbyteorder("="), copy=False)
        out = np.empty((self.hp.nClasses, H, W), np.uint8)
        if rescale and value_range is not None:
            rng = np.ascontiguousarray(np.asarray(value_range, dtype=np.uint32).reshape(C, 2))
            self._check(self._L.umx_infer_image_raw_range(self._ctx, raw.ctypes.data, raw.dtype.itemsize * 8, C, H, W, rng.ctypes.data,
                                                          float(mean), float(std), int(mode), out.ctypes.data))
            return out
        self._check(self._L.umx_infer_image_raw(self._ctx, raw.ctypes.data, raw.dtype.itemsize * 8, C, H, W,
                                                1 if rescale else 0, float(mean), float(std), int(mode), out.ctypes.data))
        return out

    def infer_image_raw_scaled(self, raw: np.ndarray, scaling: float, rescale: bool, mean: float, std: float,
                               mode: int = MODE_ACCUMULATE) -> np.ndarray:
        """The drivers' recipe at --scalingFactor != 1 on the device: raw uint8/uint16 (H,W) or (C,H,W) -> uint8 [K,H,W]."""
        raw = np.ascontiguousarray(raw)
        if raw.dtype not in (np.uint8, np.uint16):
            raise TypeError("raw planes must be uint8 or uint16")
        if raw.ndim == 2:
            raw = raw[None]
        C, H, W = raw.shape
        raw = raw.astype(raw.dtype.newbyteorder("="), copy=False)
        out = np.empty((self.hp.nClasses, H, W), np.uint8)
        self._check(self._L.umx_infer_image_raw_scaled(self._ctx, raw.ctypes.data, raw.dtype.itemsize * 8, C, H, W,
                                                       float(scaling), 1 if rescale else 0, float(mean), float(std), int(mode),
                                                       out.ctypes.data))
        return out

    def infer_image_raw_outlier(self, raw: np.ndarray, scaling: float, outlier: float, mean: float, std: float,
                                mode: int = MODE_ACCUMULATE) -> np.ndarray:
        """The drivers' recipe with --outlier on the device (any --scalingFactor): intensities rescaled to
        (min, np.percentile(resized plane, outlier)) -> (0, 0.983); raw uint8/uint16 (H,W) or (C,H,W) -> uint8 [K,H,W]."""
        raw = np.ascontiguousarray(raw)
        if raw.dtype not in (np.uint8, np.uint16):
            raise TypeError("raw planes must be uint8 or uint16")
        if raw.ndim == 2:
            raw = raw[None]
        C, H, W = raw.shape
        raw = raw.astype(raw.dtype.newbyteorder("="), copy=False)
        out = np.empty((self.hp.nClasses, H, W), np.uint8)
        self._check(self._L.umx_infer_image_raw_outlier(self._ctx, raw.ctypes.data, raw.dtype.itemsize * 8, C, H, W,
                                                        float(scaling), float(outlier), float(mean), float(std), int(mode),
                                                        out.ctypes.data))
        return out

    # -- device-pointer API (pointers are plain ints, e.g. torch.Tensor.data_ptr())
    def forward_tiles_dev(self, tiles_ptr: int, n: int, probs_ptr: int) -> None:
        self._check(self._L.umx_forward_tiles_dev(self._ctx, ctypes.c_void_p(tiles_ptr), n, ctypes.c_void_p(probs_ptr)))

    def infer_image_dev(self, image_ptr: int, C: int, H: int, W: int, mean: float, std: float, mode: int, stitch: int,
                        out_ptr: int) -> None:
        self._check(self._L.umx_infer_image_dev(self._ctx, ctypes.c_void_p(image_ptr), C, H, W, float(mean),
                                                float(std), mode, stitch, ctypes.c_void_p(out_ptr)))

    def band_tiles_dev(self, image_ptr: int, C: int, H: int, W: int, band_row0: int, band_rows: int, mean: float,
                       std: float, pr0: int, pr1: int, probs_ptr: int) -> None:
        self._check(self._L.umx_band_tiles_dev(self._ctx, ctypes.c_void_p(image_ptr), C, H, W, band_row0, band_rows,
                                               float(mean), float(std), pr0, pr1, ctypes.c_void_p(probs_ptr)))

    def stitch_dev(self, probs_ptr: int, tpr0: int, tpr1: int, H: int, W: int, mode: int, stitch: int, y0: int,
                   y1: int, out_ptr: int) -> None:
        self._check(self._L.umx_stitch_dev(self._ctx, ctypes.c_void_p(probs_ptr), tpr0, tpr1, H, W, mode, stitch, y0,
                                           y1, ctypes.c_void_p(out_ptr)))

    # -- profiling
    def profile_enable(self, on) -> None:
        """False / 0: off; True / 1: two HIP events around every launch; N >= 2: around every N-th launch of each site."""
        self._check(self._L.umx_profile_enable(self._ctx, int(on)))

    def profile_read(self) -> List[dict]:
        n = ctypes.c_int()
        arr = (ProfEntry * 256)()
        self._check(self._L.umx_profile_read(self._ctx, arr, 256, ctypes.byref(n)))
        out = []
        for i in range(min(n.value, 256)):
            e = arr[i]
            out.append({"name": e.name.decode(), "kernel": e.kernel.decode(), "launches": int(e.launches),
                        "total_ms": float(e.total_ms), "flops": float(e.flops), "bytes": float(e.bytes),
                        "exec_flops": float(e.exec_flops), "seen": int(e.launches_seen), "xcd_order": int(e.xcd_order)})
        return out
