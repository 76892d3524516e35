"""ctypes mirror of include/sgm_mi355x.h.

``SGM`` wraps the reference-shaped global entry points (SGM_Initialize / SGM_Reset / SGM_Match,
reference SemiGlobalMatching.h:78-80); ``SGMInstance`` wraps the explicit-instance extension
(several frames in flight, device-resident buffers).  Names, argument meaning and the
True/False error behaviour follow the C interface one to one.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LOAD_LOCK = threading.Lock()

STAGE_NAMES = ["census_l", "census_r", "cost", "aggr", "disp_l", "disp_r", "after_lr", "after_speckle", "final"]
_STAGE_DTYPE = [np.uint32, np.uint32, np.uint8, np.uint16] + [np.float32] * 5


class SGMOption(C.Structure):
    """Field-for-field the reference's SGMOption (SemiGlobalMatching.h:24-40), 28 bytes."""
    _fields_ = [
        ("num_paths", C.c_uint8),
        ("min_disparity", C.c_uint16),
        ("max_disparity", C.c_uint16),
        ("is_check_unique", C.c_bool),
        ("uniqueness_ratio", C.c_float),
        ("is_check_lr", C.c_bool),
        ("lrcheck_thres", C.c_float),
        ("is_remove_speckles", C.c_bool),
        ("min_speckle_area", C.c_uint16),
        ("p1", C.c_int16),
        ("p2_init", C.c_int16),
    ]


def default_option(max_disparity=64, min_disparity=0, **kw) -> SGMOption:
    """The option values the reference's driver sets (main.c:48-65), with overrides."""
    o = SGMOption()
    o.num_paths = 8
    o.min_disparity = min_disparity
    o.max_disparity = max_disparity
    o.is_check_lr = True
    o.lrcheck_thres = 1.0
    o.is_check_unique = True
    o.uniqueness_ratio = 0.99
    o.is_remove_speckles = True
    o.min_speckle_area = 50
    o.p1 = 10
    o.p2_init = 150
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def library_path() -> str:
    """The in-tree library; SGM_LIBRARY_PATH points experiments (tools/sweep_sched.sh) at another build of it."""
    return os.environ.get("SGM_LIBRARY_PATH") or os.path.join(_HERE, "libsgm_mi355x.so")


def load_library() -> C.CDLL:
    """Load libsgm_mi355x.so (built by csrc/Makefile or __graft_entry__.build()).  There is no
    fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOAD_LOCK:                                  # callers may be threads (ranks of a tile pipeline, host threads of a stream)
        if _LIB is None:
            _LIB = _load()
    return _LIB


def _load() -> C.CDLL:
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `make -C soc_project_stereo_matching_amd/csrc` "
                           "(needs hipcc); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    opt_p = C.c_void_p      # any ctypes structure with the SGMOption layout (28 bytes)
    for f in (L.SGM_Initialize, L.SGM_Reset):
        f.argtypes = [C.c_uint16, C.c_uint16, opt_p]
        f.restype = C.c_bool
    for f in (L.SGM_Match, L.SGM_MatchDevice):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = C.c_bool
    L.SGM_Synchronize.restype = C.c_bool
    L.SGM_SetDevice.argtypes = [C.c_int]
    L.SGM_SetDevice.restype = C.c_bool
    L.SGM_SetHonorNumPaths.argtypes = [C.c_int]
    L.SGM_KeepStages.argtypes = [C.c_int]
    L.SGM_ReadStage.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    L.SGM_ReadStage.restype = C.c_size_t
    L.SGM_SynthPair.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    L.SGM_Version.restype = C.c_char_p
    L.sgm_create.argtypes = [C.c_int]
    L.sgm_create.restype = C.c_void_p
    L.sgm_destroy.argtypes = [C.c_void_p]
    L.sgm_set_honor_num_paths.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_overlap_post.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_overlap_post.restype = C.c_bool
    if hasattr(L, "sgm_set_stage_cus"):       # (SGM_LIBRARY_PATH may point an A/B run at a build of older sources)
        L.sgm_set_stage_cus.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.sgm_set_stage_cus.restype = C.c_bool
        L.sgm_set_stage_priority.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sgm_set_stage_priority.restype = C.c_bool
    L.sgm_set_census_window.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_census_window.restype = C.c_bool
    L.sgm_set_reference_view.argtypes = [C.c_void_p, C.c_int]
    L.SGM_SetCensusWindow.argtypes = [C.c_int, C.c_int]
    L.SGM_SetCensusWindow.restype = C.c_bool
    L.SGM_SetReferenceView.argtypes = [C.c_int]
    L.sgm_keep_stages.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_batch.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_batch.restype = C.c_bool
    L.sgm_select_frame.argtypes = [C.c_void_p, C.c_int]
    for f in (L.sgm_initialize, L.sgm_reset):
        f.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, opt_p]
        f.restype = C.c_bool
    for f in (L.sgm_match, L.sgm_match_device, L.sgm_match_async):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = C.c_bool
    L.sgm_match_wait.argtypes = [C.c_void_p]
    L.sgm_match_wait.restype = C.c_bool
    L.sgm_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.sgm_host_alloc.restype = C.c_void_p
    L.sgm_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.sgm_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint16, opt_p, C.c_void_p]
    L.sgm_compute.restype = C.c_bool
    L.sgm_synchronize.argtypes = [C.c_void_p]
    L.sgm_synchronize.restype = C.c_bool
    L.sgm_stream.argtypes = [C.c_void_p]
    L.sgm_stream.restype = C.c_void_p
    L.sgm_read_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.sgm_read_stage.restype = C.c_size_t
    L.sgm_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_rows.restype = C.c_bool
    L.sgm_tile_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sgm_tile_begin.restype = C.c_bool
    L.sgm_tile_boundary_bytes.argtypes = [C.c_void_p]
    L.sgm_tile_boundary_bytes.restype = C.c_size_t
    for f in (L.sgm_tile_import_boundary, L.sgm_tile_export_boundary):
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        f.restype = C.c_bool
    L.sgm_tile_sweep.argtypes = [C.c_void_p, C.c_int]
    L.sgm_tile_sweep.restype = C.c_bool
    for f in (L.sgm_tile_finish, L.sgm_tile_post):
        f.argtypes = [C.c_void_p, C.c_void_p]
        f.restype = C.c_bool
    L.sgm_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.sgm_last_timing.restype = C.c_int
    L.sgm_mean_timing.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                  C.POINTER(C.c_long)]
    L.sgm_mean_timing.restype = C.c_int
    L.sgm_disparity_to_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.sgm_disparity_to_depth.restype = C.c_bool
    L.sgm_compare_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sgm_compare_depth.restype = C.c_bool
    L.sgm_gray_from_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.sgm_gray_from_planes.restype = C.c_bool
    for name in ("sgm_match_planes_async", "sgm_match_planes"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        getattr(L, name).restype = C.c_bool
    L.sgm_host_walk_line.argtypes = [C.c_int] * 5 + [C.c_void_p]
    L.sgm_host_walk_line.restype = C.c_int
    L.sgm_host_anomalous_line.argtypes = [C.c_int, C.c_int]
    L.sgm_host_anomalous_line.restype = C.c_int
    L.sgm_host_p2_table.argtypes = [C.c_int, C.c_int, C.c_void_p]
    return L


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64/libhsa; a
    second runtime (the system one libsgm_mi355x.so is linked against) initialised in the same
    process finds no GPU.  Python callers use torch for device memory and torch.distributed, so
    before loading our library we promote torch's already-loaded runtime to the global symbol
    scope: the HIP calls and the kernel registration of libsgm_mi355x.so then bind to it.  A plain C
    caller (no torch in the process) simply gets the system runtime the library is linked to."""
    try:
        import torch  # noqa: F401
    except Exception:
        return
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def synth_pair(width, height, disparity_range, seed):
    """Seeded synthetic stereo pair from the library's own host generator."""
    L = load_library()
    left = np.empty((height, width), np.uint8)
    right = np.empty((height, width), np.uint8)
    L.SGM_SynthPair(width, height, disparity_range, seed & 0xFFFFFFFF, left.ctypes.data, right.ctypes.data)
    return left, right


def _u8(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8:
        raise TypeError("images must be uint8")
    return a


class _StageReader:
    def _read(self, which):
        raise NotImplementedError

    def read_stage(self, which):
        """Copy one intermediate buffer of the last match to the host (parity tests)."""
        idx = STAGE_NAMES.index(which) if isinstance(which, str) else which
        h, w, d = self.shape
        if idx < 2 and getattr(self, "wide_census", False):
            dt, shp = np.uint64, (h, w)
        elif idx >= 10:
            dt, shp = np.uint8, (h, w, d)
        else:
            dt, shp = _STAGE_DTYPE[idx], ((h, w, d) if idx in (2, 3) else (h, w))
        out = np.empty(shp, dt)
        got = self._read(idx, out)
        if got != out.nbytes:
            raise RuntimeError(f"read_stage({which}) returned {got} of {out.nbytes} bytes")
        return out

    def read_stages(self):
        return {n: self.read_stage(n) for n in STAGE_NAMES}


class SGM(_StageReader):
    """The reference's global-instance API: SGM_Initialize / SGM_Reset / SGM_Match."""

    def __init__(self):
        self.lib = load_library()
        self.shape = None

    def set_device(self, ordinal) -> bool:
        return bool(self.lib.SGM_SetDevice(ordinal))

    def set_honor_num_paths(self, honor):
        self.lib.SGM_SetHonorNumPaths(int(honor))

    def set_census_window(self, width: int, height: int) -> bool:
        ok = bool(self.lib.SGM_SetCensusWindow(width, height))
        if ok:
            self.wide_census = not (width == 5 and height == 5)
        return ok

    def set_reference_view(self, right: bool):
        self.lib.SGM_SetReferenceView(int(right))

    def keep_stages(self, enable=True):
        self.lib.SGM_KeepStages(int(enable))

    def initialize(self, width, height, option) -> bool:
        ok = bool(self.lib.SGM_Initialize(width, height, C.byref(option)))
        if ok:
            self.shape = (height, width, option.max_disparity - option.min_disparity)
        return ok

    def reset(self, width, height, option) -> bool:
        ok = bool(self.lib.SGM_Reset(width, height, C.byref(option)))
        if ok:
            self.shape = (height, width, option.max_disparity - option.min_disparity)
        return ok

    def match(self, left, right):
        """Returns the float32 disparity map, or None where the C call returns false."""
        if left is None or right is None:
            assert not self.lib.SGM_Match(None, None, None)
            return None
        left, right = _u8(left), _u8(right)
        out = np.empty(left.shape, np.float32)
        ok = self.lib.SGM_Match(left.ctypes.data, right.ctypes.data, out.ctypes.data)
        return out if ok else None

    def compute(self, left, right, option, out=None):
        """sgm_compute: SGM_Reset + SGM_Match in one call (north_star's entry point).  None where it returns false.  `out`: a
        C-contiguous float32 [H][W] array to write into (a caller with a stream of frames allocates it once)."""
        left, right = _u8(left), _u8(right)
        h, w = left.shape
        if out is None:
            out = np.empty((h, w), np.float32)
        elif out.dtype != np.float32 or tuple(out.shape) != (h, w) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError(f"compute: out must be a C-contiguous float32 array of shape {(h, w)}")
        ok = self.lib.sgm_compute(left.ctypes.data, right.ctypes.data, w, h, C.byref(option), out.ctypes.data)
        if ok:
            self.shape = (h, w, option.max_disparity - option.min_disparity)
        return out if ok else None

    def match_device(self, d_left: int, d_right: int, d_out: int) -> bool:
        return bool(self.lib.SGM_MatchDevice(d_left, d_right, d_out))

    def synchronize(self) -> bool:
        return bool(self.lib.SGM_Synchronize())

    def shutdown(self):
        self.lib.SGM_Shutdown()

    def _read(self, idx, out):
        return self.lib.SGM_ReadStage(idx, out.ctypes.data, out.nbytes)


class SGMInstance(_StageReader):
    """Explicit instance (extension): own HIP stream and buffers on one GPU."""

    def __init__(self, device=0, batch=1):
        self.lib = load_library()
        self.handle = self.lib.sgm_create(device)
        if not self.handle:
            raise RuntimeError(f"sgm_create({device}) failed: no usable gfx950 device (no CPU fallback)")
        self.device = device
        self.shape = None
        self.batch = 1
        self._pinned = []
        if batch != 1:
            self.set_batch(batch)

    def set_batch(self, frames: int) -> bool:
        """Frames per match call ([frames][H][W] arrays); takes effect at the next initialize/reset."""
        ok = bool(self.lib.sgm_set_batch(self.handle, frames))
        if ok:
            self.batch = frames
        return ok

    def select_frame(self, frame: int):
        """Which frame of the batch read_stage() returns."""
        self.lib.sgm_select_frame(self.handle, frame)

    def close(self):
        if self.handle:
            self.lib.sgm_match_wait(self.handle)
            for p in self._pinned:
                self.lib.sgm_host_free(self.handle, p)
            self._pinned = []
            self.lib.sgm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_honor_num_paths(self, honor):
        self.lib.sgm_set_honor_num_paths(self.handle, int(honor))

    def set_overlap_post(self, enable: bool = True) -> bool:
        """The post pass (LR check, speckle, median) of a match on a second stream beside the next match's aggregation;
        results are complete after synchronize(), not in order of `stream` (include/sgm_mi355x.h)."""
        return bool(self.lib.sgm_set_overlap_post(self.handle, int(enable)))

    STAGE_MAIN, STAGE_SUM, STAGE_POST = 0, 1, 2

    def set_stage_cus(self, which: int, first_cu_per_xcd: int = 0, cus_per_xcd: int = 0) -> bool:
        """sgm_set_stage_cus: stage group `which` (STAGE_MAIN census + aggregation, STAGE_SUM cost sum + WTAs, STAGE_POST LR check +
        speckle + median) on a stream of its own restricted to CUs [first, first + count) of every XCD (count 0: all CUs,
        count < 0: back to the default)."""
        return bool(self.lib.sgm_set_stage_cus(self.handle, which, first_cu_per_xcd, cus_per_xcd))

    def set_cu_split(self, spec: str) -> bool:
        """'post=0:2,sum=2:8,main=10:22' -> set_stage_cus per group ('first:count' CUs per XCD; 'post' alone = 'post=0:0')."""
        which = {"main": self.STAGE_MAIN, "sum": self.STAGE_SUM, "post": self.STAGE_POST}
        ok = True
        for item in filter(None, (t.strip() for t in spec.split(","))):
            name, _, rng = item.partition("=")
            if rng.startswith("p"):                           # 'sum=p-1': own stream with dispatch priority -1 (higher)
                ok = bool(self.lib.sgm_set_stage_priority(self.handle, which[name], int(rng[1:]))) and ok
                continue
            first, _, count = (rng or "0:0").partition(":")
            ok = self.set_stage_cus(which[name], int(first or 0), int(count or 0)) and ok
        return ok

    def set_census_window(self, width: int, height: int) -> bool:
        """Extension: odd census window of at most 64 pixels (5x5 = reference); next initialize/reset."""
        ok = bool(self.lib.sgm_set_census_window(self.handle, width, height))
        if ok:
            self.wide_census = not (width == 5 and height == 5)
        return ok

    def set_reference_view(self, right: bool):
        """Extension: True = the result is the right image's disparity map (mirrored LR check)."""
        self.lib.sgm_set_reference_view(self.handle, int(right))

    def keep_stages(self, enable=True):
        self.lib.sgm_keep_stages(self.handle, int(enable))

    def fused_sweep_rows(self) -> int:
        """Rows per workgroup of the fused last sweep in the LAST match, 0 if it ran the separate kernels."""
        self.lib.sgm_fused_sweep_rows.argtypes = [C.c_void_p]
        self.lib.sgm_fused_sweep_rows.restype = C.c_int
        return int(self.lib.sgm_fused_sweep_rows(self.handle))

    def enable_timing(self, enable=True):
        self.lib.sgm_enable_timing(self.handle, int(enable))

    def last_timing(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = self.lib.sgm_last_timing(self.handle, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def mean_timing(self):
        """({stage: mean ms}, {stage: min ms}, matches) over every match since enable_timing()."""
        names = (C.c_char_p * 16)()
        mean = (C.c_float * 16)()
        mn = (C.c_float * 16)()
        cnt = C.c_long(0)
        n = self.lib.sgm_mean_timing(self.handle, names, mean, mn, 16, C.byref(cnt))
        return ({names[i].decode(): float(mean[i]) for i in range(n)},
                {names[i].decode(): float(mn[i]) for i in range(n)}, int(cnt.value))

    def initialize(self, width, height, option) -> bool:
        ok = bool(self.lib.sgm_initialize(self.handle, width, height, C.byref(option)))
        if ok:
            self.shape = (height, width, option.max_disparity - option.min_disparity)
        return ok

    def reset(self, width, height, option) -> bool:
        ok = bool(self.lib.sgm_reset(self.handle, width, height, C.byref(option)))
        if ok:
            self.shape = (height, width, option.max_disparity - option.min_disparity)
        return ok

    def match(self, left, right):
        """left/right: uint8 [H][W], or [batch][H][W] when the instance has a batch > 1."""
        if self.shape is None:
            return None                                   # Match before Initialize: false in the reference (.c:70)
        left, right = _u8(left), _u8(right)
        want = self.shape[:2] if self.batch == 1 else (self.batch,) + tuple(self.shape[:2])
        if tuple(left.shape) != tuple(want):
            raise ValueError(f"expected images of shape {want}, got {left.shape}")
        out = np.empty(left.shape, np.float32)
        ok = self.lib.sgm_match(self.handle, left.ctypes.data, right.ctypes.data, out.ctypes.data)
        return out if ok else None

    def match_async(self, left, right, out) -> bool:
        """sgm_match_async: queue upload + pipeline + download and return.  `left`, `right` (uint8) and `out` (float32)
        must be C-contiguous arrays of the instance's shape that stay alive and untouched until match_wait()."""
        if self.shape is None:
            return False                                  # Match before Initialize: false in the reference (.c:70)
        for a in (left, right, out):
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("match_async needs C-contiguous arrays")
        if left.dtype != np.uint8 or right.dtype != np.uint8 or out.dtype != np.float32:
            raise TypeError("match_async: uint8 images, float32 output")
        # the C side moves batch * W * H bytes (4x that for the output) whatever the arrays hold: check before handing pointers over
        want = self._frame_shape()
        for name, a in (("left", left), ("right", right), ("out", out)):
            if tuple(a.shape) != want:
                raise ValueError(f"match_async: {name} has shape {tuple(a.shape)}, the instance expects {want}")
        return bool(self.lib.sgm_match_async(self.handle, left.ctypes.data, right.ctypes.data, out.ctypes.data))

    def _frame_shape(self):
        h, w = self.shape[:2]
        return (h, w) if self.batch == 1 else (self.batch, h, w)

    def match_wait(self) -> bool:
        return bool(self.lib.sgm_match_wait(self.handle))

    def host_array(self, shape, dtype):
        """A numpy array in page-locked host memory (sgm_host_alloc): sgm_match_async uses it in place, without the
        staging copy.  Freed when the instance is closed; do not use it after that."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.sgm_host_alloc(self.handle, n)
        if not p:
            raise MemoryError("sgm_host_alloc failed")
        self._pinned.append(p)
        buf = (C.c_uint8 * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def match_device(self, d_left: int, d_right: int, d_out: int) -> bool:
        """Device pointers (e.g. torch tensor .data_ptr()); asynchronous on the instance stream."""
        return bool(self.lib.sgm_match_device(self.handle, d_left, d_right, d_out))

    def synchronize(self) -> bool:
        return bool(self.lib.sgm_synchronize(self.handle))

    # ---- row tiles (one frame over several GPUs; include/sgm_mi355x.h "row tiles"); all device pointers ----
    def set_rows(self, row_begin: int, row_end: int) -> bool:
        """This instance computes rows [row_begin, row_end) from the next initialize/reset on; (0, 0) = whole frames."""
        return bool(self.lib.sgm_set_rows(self.handle, row_begin, row_end))

    def tile_begin(self, d_left: int, d_right: int) -> bool:
        return bool(self.lib.sgm_tile_begin(self.handle, d_left, d_right))

    def tile_boundary_bytes(self) -> int:
        return int(self.lib.sgm_tile_boundary_bytes(self.handle))

    def tile_import_boundary(self, forward: bool, d_buf: int) -> bool:
        return bool(self.lib.sgm_tile_import_boundary(self.handle, int(forward), d_buf))

    def tile_sweep(self, forward: bool) -> bool:
        return bool(self.lib.sgm_tile_sweep(self.handle, int(forward)))

    def tile_export_boundary(self, forward: bool, d_buf: int) -> bool:
        return bool(self.lib.sgm_tile_export_boundary(self.handle, int(forward), d_buf))

    def tile_finish(self, d_disp: int) -> bool:
        return bool(self.lib.sgm_tile_finish(self.handle, d_disp))

    def tile_post(self, d_disp: int) -> bool:
        return bool(self.lib.sgm_tile_post(self.handle, d_disp))

    # ---- test-platform arithmetic on device buffers (include/sgm_mi355x.h) ----
    def disparity_to_depth(self, d_disp: int, count: int, fx: float, baseline: float, doffs: float, d_depth: int) -> bool:
        return bool(self.lib.sgm_disparity_to_depth(self.handle, d_disp, count, fx, baseline, doffs, d_depth))

    def compare_depth(self, d_ground_truth: int, d_test: int, count: int, abs_thresh: float = 10.0):
        """(rmse, bad_pixel_rate, n_valid) of two device depth images; None where the C call returns false."""
        rmse, bpr, n = C.c_double(), C.c_double(), C.c_uint64()
        ok = self.lib.sgm_compare_depth(self.handle, d_ground_truth, d_test, count, abs_thresh, C.byref(rmse), C.byref(bpr), C.byref(n))
        return (rmse.value, bpr.value, int(n.value)) if ok else None

    # ---- a test-platform frame end to end (include/sgm_mi355x.h) ----
    def gray_from_planes(self, d_bgr: int, count: int, d_gray: int, weight_r: int = 76) -> bool:
        """Device pointers: three `count`-byte planes B, G, R -> grey bytes; asynchronous on the instance stream."""
        return bool(self.lib.sgm_gray_from_planes(self.handle, d_bgr, count, weight_r, d_gray))

    def match_planes(self, planes, fx: float, baseline: float, doffs: float, depth, wait: bool = True) -> bool:
        """sgm_match_planes(_async): `planes` uint8 [batch *] 6 x H x W (left B, G, R, right B, G, R), `depth` float32 H x W per
        frame; with wait=False the arrays must stay alive and untouched until match_wait()."""
        for a in (planes, depth):
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("match_planes needs C-contiguous arrays")
        if planes.dtype != np.uint8 or depth.dtype != np.float32:
            raise TypeError("match_planes: uint8 planes, float32 depth")
        if self.shape is None:
            return False
        h, w = self.shape[:2]
        if planes.size != self.batch * 6 * h * w or tuple(planes.shape[-2:]) != (h, w):
            raise ValueError(f"match_planes: planes of shape {tuple(planes.shape)}, expected [{self.batch} x] 6 x {h} x {w}")
        if depth.size != self.batch * h * w or tuple(depth.shape[-2:]) != (h, w):
            raise ValueError(f"match_planes: depth of shape {tuple(depth.shape)}, expected [{self.batch} x] {h} x {w}")
        fn = self.lib.sgm_match_planes if wait else self.lib.sgm_match_planes_async
        return bool(fn(self.handle, planes.ctypes.data, fx, baseline, doffs, depth.ctypes.data))

    @property
    def stream(self) -> int:
        return self.lib.sgm_stream(self.handle) or 0

    def _read(self, idx, out):
        return self.lib.sgm_read_stage(self.handle, idx, out.ctypes.data, out.nbytes)
