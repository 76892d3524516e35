"""ctypes bindings for the checker libraries -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/sgm_oracle.h).  Two libraries are wrapped:

* ``Oracle``  -- oracle/libsgm_oracle.so, our CPU restatement (always buildable, gcc only);
* ``Reference`` -- oracle/_ref/libsgm_ref_<W>x<H>x<D>.so, the reference's own C compiled by
  oracle/build_ref.sh (present only if it was built where /root/reference exists).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INF = np.float32(np.inf)

# reference direction order, SemiGlobalMatching.c:213-220
DIRECTIONS = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]


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
    """The options main.c:48-65 sets, with overrides."""
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


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def build_oracle(force=False) -> str:
    path = os.path.join(HERE, "libsgm_oracle.so")
    src = os.path.join(HERE, "sgm_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "libsgm_oracle.so"])
    return path


STAGE_NAMES = ["census_l", "census_r", "cost", "aggr", "disp_l", "disp_r", "after_lr", "after_speckle", "final"]
_STAGE_DT = [np.uint32, np.uint32, np.uint8, np.uint16] + [np.float32] * 5


class Oracle:
    def __init__(self, path=None):
        self.lib = L = C.CDLL(path or build_oracle())
        L.sgmo_create.restype = C.c_void_p
        L.sgmo_destroy.argtypes = [C.c_void_p]
        L.sgmo_set_honor_num_paths.argtypes = [C.c_void_p, C.c_int]
        L.sgmo_set_census_window.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sgmo_set_census_window.restype = C.c_bool
        L.sgmo_set_reference_view.argtypes = [C.c_void_p, C.c_int]
        L.sgmo_census_window.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        for f in (L.sgmo_initialize, L.sgmo_reset):
            f.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.POINTER(SGMOption)]
            f.restype = C.c_bool
        L.sgmo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sgmo_match.restype = C.c_bool
        L.sgmo_clear_census.argtypes = [C.c_void_p]
        L.sgmo_stage.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.sgmo_stage.restype = C.c_void_p
        L.sgmo_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.sgmo_path_walk.argtypes = [C.c_int] * 5 + [C.c_void_p]
        L.sgmo_path_walk.restype = C.c_int
        L.sgmo_path_lines.argtypes = [C.c_int] * 4
        L.sgmo_path_lines.restype = C.c_int
        L.sgmo_synth_pair.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.sgmo_census5x5.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.sgmo_cost.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        L.sgmo_aggregate_dir.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 3
        L.sgmo_aggregate_all.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
        L.sgmo_wta.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_bool, C.c_float, C.c_int, C.c_void_p]
        L.sgmo_lrcheck.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.sgmo_remove_speckles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint]
        L.sgmo_median3_inplace.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sgmo_normalize_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.ctx = C.c_void_p(L.sgmo_create())
        self.shape = None

    def __del__(self):
        try:
            self.lib.sgmo_destroy(self.ctx)
        except Exception:
            pass

    # --- reference-shaped API -------------------------------------------------
    def set_honor_num_paths(self, honor: bool):
        self.lib.sgmo_set_honor_num_paths(self.ctx, int(honor))

    def set_census_window(self, cw, ch) -> bool:
        ok = bool(self.lib.sgmo_set_census_window(self.ctx, cw, ch))
        if ok:
            self.wide_census = not (cw == 5 and ch == 5)
        return ok

    def set_reference_view(self, right: bool):
        self.lib.sgmo_set_reference_view(self.ctx, int(right))

    def census_window(self, img, cw, ch):
        h, w = img.shape
        out = np.empty((h, w), np.uint64)
        self.lib.sgmo_census_window(np.ascontiguousarray(img).ctypes.data, w, h, cw, ch, out.ctypes.data)
        return out

    def initialize(self, w, h, opt) -> bool:
        ok = self.lib.sgmo_initialize(self.ctx, w, h, C.byref(opt))
        self.shape = (h, w, opt.max_disparity - opt.min_disparity)
        return ok

    def reset(self, w, h, opt) -> bool:
        ok = self.lib.sgmo_reset(self.ctx, w, h, C.byref(opt))
        self.shape = (h, w, opt.max_disparity - opt.min_disparity)
        return ok

    def match(self, left, right):
        if left is None or right is None:
            return None if not self.lib.sgmo_match(self.ctx, None, None, None) else None
        h, w, _ = self.shape
        out = np.empty((h, w), np.float32)
        ok = self.lib.sgmo_match(self.ctx, left.ctypes.data, right.ctypes.data, out.ctypes.data)
        return out if ok else None

    def stage(self, which):
        if isinstance(which, str):
            which = STAGE_NAMES.index(which)
        n = C.c_size_t()
        ptr = self.lib.sgmo_stage(self.ctx, which, C.byref(n))
        buf = (C.c_uint8 * n.value).from_address(ptr)
        dt = np.uint64 if (which < 2 and getattr(self, "wide_census", False)) else _STAGE_DT[which]
        a = np.frombuffer(buf, dtype=dt).copy()
        h, w, d = self.shape
        return a.reshape((h, w, d) if which in (2, 3) else (h, w))

    def stages(self):
        return {n: self.stage(i) for i, n in enumerate(STAGE_NAMES)}

    def counters(self):
        c = (C.c_uint64 * 2)()
        self.lib.sgmo_counters(self.ctx, c)
        return {"oob_dropped": int(c[0]), "u8_wraps": int(c[1])}

    def clear_census(self):
        """The census statics as a new process finds them (all zero): the reset()/match() sequence API otherwise keeps, like
        the reference, what earlier frames of other shapes left in the words census_transform_5x5 never writes (Q3)."""
        self.lib.sgmo_clear_census(self.ctx)

    def run(self, left, right, opt):
        """One frame as a fresh process computes it: clear the census statics, Reset + Match; returns the dict of all nine stages."""
        h, w = left.shape
        self.clear_census()
        assert self.reset(w, h, opt)
        assert self.match(np.ascontiguousarray(left), np.ascontiguousarray(right)) is not None
        return self.stages()

    # --- single stages --------------------------------------------------------
    def path_walk(self, w, h, dx, dy, line):
        pix = np.empty(max(w, h), np.int32)
        n = self.lib.sgmo_path_walk(w, h, dx, dy, line, pix.ctypes.data)
        return pix[:n].copy()

    def path_lines(self, w, h, dx, dy):
        return self.lib.sgmo_path_lines(w, h, dx, dy)

    def synth_pair(self, w, h, d, seed):
        l = np.empty((h, w), np.uint8)
        r = np.empty((h, w), np.uint8)
        self.lib.sgmo_synth_pair(w, h, d, seed & 0xFFFFFFFF, l.ctypes.data, r.ctypes.data)
        return l, r

    def census(self, img):
        h, w = img.shape
        out = np.empty((h, w), np.uint32)
        self.lib.sgmo_census5x5(np.ascontiguousarray(img).ctypes.data, w, h, out.ctypes.data)
        return out

    def cost(self, cl, cr, dmin, dmax):
        h, w = cl.shape
        out = np.empty((h, w, dmax - dmin), np.uint8)
        self.lib.sgmo_cost(cl.ctypes.data, cr.ctypes.data, w, h, dmin, dmax, out.ctypes.data)
        return out

    def aggregate_dir(self, img, cost, p1, p2, dx, dy, S=None, want_last=False, want_visits=False):
        h, w, d = cost.shape
        if S is None:
            S = np.zeros((h, w, d), np.uint16)
        last = np.zeros((h, w, d), np.uint8) if want_last else None
        vis = np.zeros((h, w), np.uint8) if want_visits else None
        self.lib.sgmo_aggregate_dir(np.ascontiguousarray(img).ctypes.data, cost.ctypes.data, w, h, d, p1, p2, dx, dy,
                                    S.ctypes.data, last.ctypes.data if want_last else None,
                                    vis.ctypes.data if want_visits else None)
        return S, last, vis

    def aggregate_all(self, img, cost, p1, p2, n_dirs=8):
        h, w, d = cost.shape
        S = np.zeros((h, w, d), np.uint16)
        self.lib.sgmo_aggregate_all(np.ascontiguousarray(img).ctypes.data, cost.ctypes.data, w, h, d, p1, p2, n_dirs,
                                    S.ctypes.data)
        return S

    def wta(self, S, dmin, dmax, unique, ratio, right_view):
        h, w, _ = S.shape
        out = np.empty((h, w), np.float32)
        self.lib.sgmo_wta(S.ctypes.data, w, h, dmin, dmax, unique, ratio, int(right_view), out.ctypes.data)
        return out

    def lrcheck(self, dl, dr, thres):
        dl = dl.copy()
        h, w = dl.shape
        self.lib.sgmo_lrcheck(dl.ctypes.data, np.ascontiguousarray(dr).ctypes.data, w, h, thres)
        return dl

    def remove_speckles(self, d, min_area, diff=1.0):
        d = d.copy()
        h, w = d.shape
        self.lib.sgmo_remove_speckles(d.ctypes.data, w, h, diff, min_area)
        return d

    def median(self, d):
        d = d.copy()
        h, w = d.shape
        self.lib.sgmo_median3_inplace(d.ctypes.data, w, h)
        return d

    def normalize_u8(self, d):
        h, w = d.shape
        out = np.empty((h, w), np.uint8)
        self.lib.sgmo_normalize_u8(np.ascontiguousarray(d).ctypes.data, w, h, out.ctypes.data)
        return out


def ref_path(w, h, d):
    """Smallest built reference library whose capacity covers (w, h, d), or None."""
    refdir = os.path.join(HERE, "_ref")
    best = None
    if os.path.isdir(refdir):
        for f in os.listdir(refdir):
            if f.startswith("libsgm_ref_") and f.endswith(".so"):
                cw, ch, cd = (int(v) for v in f[len("libsgm_ref_"):-3].split("x"))
                if cw >= w and ch >= h and cd >= d and cw * ch >= w * h:
                    size = cw * ch * cd
                    if best is None or size < best[0]:
                        best = (size, os.path.join(refdir, f))
    return best[1] if best else None


class Reference:
    """The reference's own C (guarded build, see oracle/build_ref.sh)."""

    def __init__(self, path):
        self.path = path
        self.lib = L = C.CDLL(path)
        for f in (L.SGM_Initialize, L.SGM_Reset):
            f.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(SGMOption)]
            f.restype = C.c_bool
        L.SGM_Match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.SGM_Match.restype = C.c_bool
        L.ref_run_stages.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint16, C.POINTER(SGMOption)] + [C.c_void_p] * 9
        L.ref_run_stages.restype = C.c_int
        if hasattr(L, "ref_run_stages_first_dirs"):
            L.ref_run_stages_first_dirs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint16, C.POINTER(SGMOption), C.c_int] + [C.c_void_p] * 9
            L.ref_run_stages_first_dirs.restype = C.c_int
        L.ref_aggregate_dir.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint16, C.POINTER(SGMOption), C.c_int, C.c_int, C.c_void_p]
        L.ref_aggregate_dir.restype = C.c_int
        L.ref_oob_count.restype = C.c_ulong
        cap = (C.c_int * 3)()
        L.ref_capacity(cap)
        self.capacity = tuple(cap)

    @classmethod
    def for_shape(cls, w, h, d):
        p = ref_path(w, h, d)
        return cls(p) if p else None

    def run(self, left, right, opt, first_dirs=None):
        """All nine stages by the reference's own functions.  first_dirs = n: only the first n of the reference's eight CostAggregate
        calls (SemiGlobalMatching.c:213-220) -- the 4-path mode as SURVEY.md Q1 defines it."""
        h, w = left.shape
        d = opt.max_disparity - opt.min_disparity
        out = {
            "census_l": np.zeros((h, w), np.uint32), "census_r": np.zeros((h, w), np.uint32),
            "cost": np.zeros((h, w, d), np.uint8), "aggr": np.zeros((h, w, d), np.uint16),
        }
        for n in STAGE_NAMES[4:]:
            out[n] = np.zeros((h, w), np.float32)
        left = np.ascontiguousarray(left)
        right = np.ascontiguousarray(right)
        if first_dirs is None:
            rc = self.lib.ref_run_stages(left.ctypes.data, right.ctypes.data, w, h, C.byref(opt),
                                         *[out[n].ctypes.data for n in STAGE_NAMES])
        else:
            rc = self.lib.ref_run_stages_first_dirs(left.ctypes.data, right.ctypes.data, w, h, C.byref(opt), int(first_dirs),
                                                    *[out[n].ctypes.data for n in STAGE_NAMES])
        if rc != 0:
            raise RuntimeError(f"ref_run_stages rc={rc}")
        if not opt.is_check_lr:
            out["disp_r"][:] = 0
        return out

    def aggregate_dir(self, img, cost, opt, dx, dy):
        h, w, d = cost.shape
        S = np.zeros((h, w, d), np.uint16)
        rc = self.lib.ref_aggregate_dir(np.ascontiguousarray(img).ctypes.data, np.ascontiguousarray(cost).ctypes.data,
                                        w, h, C.byref(opt), dx, dy, S.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"ref_aggregate_dir rc={rc}")
        return S

    def oob_count(self):
        return int(self.lib.ref_oob_count())

    def clear_census(self):
        self.lib.ref_clear_census()

    def api_match(self, left, right, opt, reset=True, clear=True):
        """Through the reference's public entry points only (what bench's cpu_baseline times).  clear=False leaves the reference's
        static census buffers as the calls before left them (the raw call sequence of a long-running process, Q3)."""
        h, w = left.shape
        if reset:
            if clear:
                self.lib.ref_clear_census()     # Q3: stale border values from an earlier shape
            if not self.lib.SGM_Reset(w, h, C.byref(opt)):
                return None
        out = np.empty((h, w), np.float32)
        ok = self.lib.SGM_Match(left.ctypes.data, right.ctypes.data, out.ctypes.data)
        return out if ok else None


def load_gray_stb(path) -> np.ndarray:
    """Decode an RGB/grey PNG to the 8-bit grey stb_image produces for req_comp=1
    (stb_image.h:1746-1749: (77 r + 150 g + 29 b) >> 8)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode == "L":
        return np.asarray(im, np.uint8).copy()
    a = np.asarray(im.convert("RGB")).astype(np.uint32)
    return ((a[..., 0] * 77 + a[..., 1] * 150 + a[..., 2] * 29) >> 8).astype(np.uint8)
