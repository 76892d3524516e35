"""ctypes mirror of include/sgm_tiles.h: the C host of the multi-GPU row-tile path.

Three layers, as in the header:

* ``tile_step`` -- the step schedule (csrc/sgm_tile_sched.c) driven with a Python engine (``PyEngine``): what
  tiling.TilePipeline and its CPU tests over gloo run;
* ``TilesPipeline`` -- the device pipeline of one rank (csrc/sgm_tiles.c: slots, streams, events, hand-over buffers), no
  Python on the path between ``submit`` calls;
* transports -- ``rccl_transport`` (RCCL over xGMI, the product), ``LocalGroup`` (ranks as threads of one process on one
  GPU), ``NullTransport`` (moves nothing: one rank's share of the work alone on its GPU).

The reference has no multi-GPU code; the call site this serves is a C caller with a stream of frames
(ZedBoard/Vitis/lwip_tcp_perf_client/src/stereo_matching.c:34-40).  Everything here only forwards to the C library.
"""
from __future__ import annotations

import ctypes as C
import threading

from .sgm import load_library

XOP_SEND, XOP_RECV = 0, 1
XBUF_BOUNDARY, XBUF_ROWS = 0, 1
ID_BYTES = 128


class TileXop(C.Structure):
    """sgm_tile_xop: one operation of a step's grouped exchange."""
    _fields_ = [("kind", C.c_int), ("buf", C.c_int), ("slot", C.c_int), ("forward", C.c_int), ("incoming", C.c_int),
                ("row_begin", C.c_int), ("row_end", C.c_int), ("peer", C.c_int)]


_FN_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_long)
_FN_SLOT_DIR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
_FN_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TileXop), C.c_int, C.POINTER(C.c_int), C.c_int)
_FN_SLOT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
_FN_POST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_long)


class TileEngine(C.Structure):
    """sgm_tile_engine: what a rank's engine offers to sgm_tile_step."""
    _fields_ = [("user", C.c_void_p), ("begin", _FN_BEGIN), ("import_boundary", _FN_SLOT_DIR), ("sweep", _FN_SLOT_DIR),
                ("export_boundary", _FN_SLOT_DIR), ("exchange", _FN_EXCHANGE), ("finish", _FN_SLOT), ("post", _FN_POST)]


_FN_GROUP = C.CFUNCTYPE(C.c_int, C.c_void_p)
_FN_XFER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
_FN_DESTROY = C.CFUNCTYPE(None, C.c_void_p)


class Transport(C.Structure):
    """sgm_tiles_transport: grouped point-to-point operations on a HIP stream."""
    _fields_ = [("ctx", C.c_void_p), ("group_start", _FN_GROUP), ("send", _FN_XFER), ("recv", _FN_XFER),
                ("group_end", _FN_GROUP), ("destroy", _FN_DESTROY)]

    def close(self):
        if self.destroy and self.ctx:
            self.destroy(self.ctx)
            self.ctx = None


_RESULT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p)
_BOUND = False
_BIND_LOCK = threading.Lock()


def lib():
    """The library with the sgm_tiles.h prototypes attached."""
    L = load_library()
    if _BOUND:
        return L
    # ranks may be threads of one process: the first use must not race.  (ctypes creates a function object per first attribute
    # access; two threads binding at once can leave the cached object without prototypes, and a `long` or a pointer then goes
    # through as a 32-bit int.)
    with _BIND_LOCK:
        if not _BOUND:
            _bind(L)
    return L


def _bind(L):
    global _BOUND
    i, p = C.c_int, C.c_void_p
    L.sgm_tile_rows.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    L.sgm_tile_rows.restype = C.c_bool
    L.sgm_tile_slots_needed.argtypes = [i, i]
    L.sgm_tile_slots_needed.restype = i
    L.sgm_tile_slot_bytes.argtypes = [i, i, C.c_uint16, C.c_uint16, p, i]
    L.sgm_tile_slot_bytes.restype = C.c_size_t
    L.sgm_tile_steps_total.argtypes = [C.c_long, i, i]
    L.sgm_tile_steps_total.restype = C.c_long
    L.sgm_tile_step.argtypes = [C.POINTER(TileEngine), i, i, i, i, i, C.c_long, C.c_long]
    L.sgm_tile_step.restype = i
    L.sgm_tiles_rccl_unique_id.argtypes = [p]
    L.sgm_tiles_rccl_unique_id.restype = C.c_bool
    L.sgm_tiles_rccl_transport.argtypes = [p, i, i, i, C.POINTER(Transport)]
    L.sgm_tiles_rccl_transport.restype = C.c_bool
    L.sgm_tiles_local_group.argtypes = [i, i]
    L.sgm_tiles_local_group.restype = p
    L.sgm_tiles_local_transport.argtypes = [p, i, C.POINTER(Transport)]
    L.sgm_tiles_local_transport.restype = C.c_bool
    L.sgm_tiles_local_destroy.argtypes = [p]
    L.sgm_tiles_create.argtypes = [i, i, i, C.c_uint16, C.c_uint16, p, i, i, i, i, C.POINTER(Transport)]
    L.sgm_tiles_create.restype = p
    L.sgm_tiles_destroy.argtypes = [p]
    L.sgm_tiles_set_honor_num_paths.argtypes = [p, i]
    L.sgm_tiles_on_result.argtypes = [p, _RESULT_FN, p]
    L.sgm_tiles_result_ring.argtypes = [p, p, i]
    L.sgm_tiles_submit.argtypes = [p, p, p, p]
    L.sgm_tiles_submit.restype = C.c_bool
    L.sgm_tiles_finish.argtypes = [p]
    L.sgm_tiles_finish.restype = C.c_bool
    L.sgm_tiles_info.argtypes = [p, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    _BOUND = True
    return L


def tile_rows(height: int, world: int, rank: int):
    """sgm_tile_rows: rows [r0, r1) of `rank`; ValueError where the C function returns false."""
    r0, r1 = C.c_int(), C.c_int()
    if not lib().sgm_tile_rows(height, world, rank, C.byref(r0), C.byref(r1)):
        raise ValueError(f"cannot cut {height} rows into {world} tiles")
    return r0.value, r1.value


def slot_bytes(row_begin: int, row_end: int, width: int, height: int, option, batch: int = 1) -> int:
    """Device memory one slot of a rank owning rows [row_begin, row_end) takes (include/sgm_tiles.h: sgm_tile_slot_bytes)."""
    return int(lib().sgm_tile_slot_bytes(row_begin, row_end, width, height, C.byref(option), batch))


def slots_needed(world: int, lead: int = 0) -> int:
    return int(lib().sgm_tile_slots_needed(world, lead))


def steps_total(n_frames: int, world: int, lead: int = 0) -> int:
    return int(lib().sgm_tile_steps_total(n_frames, world, lead))


class PyEngine:
    """An sgm_tile_engine whose functions are Python callables:
         begin(slot, frame), import_boundary(slot, forward), sweep(slot, forward), export_boundary(slot, forward),
         exchange(ops, slots)  -- ops: list of TileXop copies, slots: sorted list of the slots they touch --,
         finish(slot), post(slot, frame).
    A callable that raises ends the step; `step` re-raises the exception on the Python side."""

    def __init__(self, begin, import_boundary, sweep, export_boundary, exchange, finish, post):
        self.error = None

        def guard(fn):
            def call(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as e:                   # must not propagate through the C frames
                    self.error = e
                    return -100
            return call

        def xch(_user, ops, n_ops, slots, n_slots):
            copies = []
            for k in range(n_ops):
                o = TileXop()
                C.memmove(C.byref(o), C.byref(ops[k]), C.sizeof(TileXop))
                copies.append(o)
            return guard(exchange)(copies, [int(slots[k]) for k in range(n_slots)])

        self._keep = (
            _FN_BEGIN(lambda _u, slot, frame: guard(begin)(slot, frame)),
            _FN_SLOT_DIR(lambda _u, slot, fwd: guard(import_boundary)(slot, bool(fwd))),
            _FN_SLOT_DIR(lambda _u, slot, fwd: guard(sweep)(slot, bool(fwd))),
            _FN_SLOT_DIR(lambda _u, slot, fwd: guard(export_boundary)(slot, bool(fwd))),
            _FN_EXCHANGE(xch),
            _FN_SLOT(lambda _u, slot: guard(finish)(slot)),
            _FN_POST(lambda _u, slot, frame: guard(post)(slot, frame)),
        )
        self.struct = TileEngine(None, *self._keep)

    def step(self, rank: int, world: int, height: int, slots: int, lead: int, step: int, frames_known: int):
        """sgm_tile_step; raises what an engine function raised, RuntimeError for any other non-zero return."""
        self.error = None
        rc = lib().sgm_tile_step(C.byref(self.struct), rank, world, height, slots, lead, step, frames_known)
        if self.error is not None:
            err, self.error = self.error, None
            raise err
        if rc != 0:
            raise RuntimeError(f"sgm_tile_step returned {rc} (rank {rank} of {world}, step {step}, {slots} slots, lead {lead})")


def rccl_unique_id() -> bytes:
    """sgm_tiles_rccl_unique_id on rank 0; move the bytes to every rank, then rccl_transport."""
    buf = C.create_string_buffer(ID_BYTES)
    if not lib().sgm_tiles_rccl_unique_id(buf):
        raise RuntimeError("sgm_tiles_rccl_unique_id failed (RCCL not available?)")
    return buf.raw


def rccl_transport(uid: bytes, rank: int, world: int, device: int) -> Transport:
    if len(uid) != ID_BYTES:
        raise ValueError(f"an RCCL id has {ID_BYTES} bytes")
    t = Transport()
    if not lib().sgm_tiles_rccl_transport(C.create_string_buffer(uid, ID_BYTES), rank, world, device, C.byref(t)):
        raise RuntimeError(f"sgm_tiles_rccl_transport failed for rank {rank} of {world} on device {device}")
    return t


class LocalGroup:
    """sgm_tiles_local_*: `world` ranks as threads of this process on one GPU."""

    def __init__(self, world: int, device: int = 0):
        self.handle = lib().sgm_tiles_local_group(world, device)
        if not self.handle:
            raise RuntimeError("sgm_tiles_local_group failed")
        self.world = world

    def transport(self, rank: int) -> Transport:
        t = Transport()
        if not lib().sgm_tiles_local_transport(self.handle, rank, C.byref(t)):
            raise RuntimeError("sgm_tiles_local_transport failed")
        return t

    def close(self):
        if self.handle:
            lib().sgm_tiles_local_destroy(self.handle)
            self.handle = None


class NullTransport:
    """A transport whose operations move nothing (tools/tiles_schedule_cost.py: one rank of N alone on its GPU; the results
    are meaningless, the launches are exactly that rank's share)."""

    def __init__(self):
        self._keep = (_FN_GROUP(lambda c: 0), _FN_XFER(lambda c, b, n, peer, st: 0), _FN_XFER(lambda c, b, n, peer, st: 0),
                      _FN_GROUP(lambda c: 0))
        self.struct = Transport(None, self._keep[0], self._keep[1], self._keep[2], self._keep[3], _FN_DESTROY())


class TilesPipeline:
    """sgm_tiles_*: the row-tile pipeline of one rank on its GPU.  `transport`: a Transport (or NullTransport().struct) when
    world > 1.  Frames are [batch][H][W] device arrays; every rank submits the same frames in the same order."""

    def __init__(self, device: int, rank: int, world: int, width: int, height: int, option, batch: int = 1, lead: int = 2,
                 spare: int = 1, throttle: int = 0, transport=None, honor_num_paths: bool = False):
        L = lib()
        self.lib = L
        self._transport = transport
        tp = C.byref(transport) if transport is not None else None
        self.handle = L.sgm_tiles_create(device, rank, world, width, height, C.byref(option), batch, lead, spare, throttle, tp)
        if not self.handle:
            raise RuntimeError(f"sgm_tiles_create failed (rank {rank} of {world}, {width}x{height}, batch {batch})")
        if honor_num_paths:
            L.sgm_tiles_set_honor_num_paths(self.handle, 1)
        self.rank, self.world, self.batch, self.shape = rank, world, batch, (height, width)
        self._cb = None

    def info(self):
        """(r0, r1, slots)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.lib.sgm_tiles_info(self.handle, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def on_result(self, fn):
        """fn(frame, d_map_pointer, hip_event_pointer) on the owner rank, from inside submit / finish."""
        self._cb = _RESULT_FN(lambda _u, frame, d_map, ev: fn(int(frame), d_map, ev))
        self.lib.sgm_tiles_on_result(self.handle, self._cb, None)

    def result_ring(self, d_ring: int, ring_frames: int):
        self.lib.sgm_tiles_result_ring(self.handle, d_ring, ring_frames)

    def submit(self, d_left: int, d_right: int, ready_event: int = 0) -> bool:
        return bool(self.lib.sgm_tiles_submit(self.handle, d_left, d_right, ready_event or None))

    def finish(self) -> bool:
        return bool(self.lib.sgm_tiles_finish(self.handle))

    def close(self):
        if self.handle:
            self.lib.sgm_tiles_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
