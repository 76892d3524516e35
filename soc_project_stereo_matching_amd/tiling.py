"""Row tiles: ONE frame spread over the GPUs of a node (one process per GPU).

Frame-level sharding (sharding.py) is the throughput path and needs no communication.  Row tiling is the
other way north_star allows the path to shard: rank r computes rows [r0, r1) of every frame.  Horizontal
paths, the cost sum, both winner-take-all passes and the LR check are row-local; the vertical and diagonal
paths cross the tile borders, so each of the two vertical sweeps is a chain over the ranks that hands over
one image row of path costs per direction (3 x W x Dp bytes, 477 KB at KITTI size with D = 128):

    forward sweep  (directions with dy = +1):  rank 0 -> 1 -> ... -> N-1
    backward sweep (directions with dy = -1):  rank N-1 -> ... -> 0

The two chains run in opposite directions at the same time (ranks in the upper half of the frame do the
forward sweep first, the others the backward sweep first), each on its own process group so that a rank's
send on one chain never queues behind a receive of the other.  The per-tile disparity rows are then
all-gathered (W x H x 4 bytes in total) and every rank runs speckle removal and the median on the whole
map -- both are whole-frame passes (connected components, a raster-order recurrence).

`TileEngine` is what a rank drives; `DeviceTileEngine` is the product (an SGMInstance on the rank's GPU,
torch tensors for device memory).  The schedule below is independent of the engine, which is how the CPU
test exercises it over gloo with a toy engine.  The serial chain is inherent to the algorithm (a vertical
path is H dependent steps), so row tiling does not shorten one frame's aggregation; it spreads the
horizontal paths, the cost sum and the WTAs, and the memory of very large frames.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def tile_rows(height: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced row ranges, one per rank (every rank gets at least one row)."""
    if world < 1 or height < world:
        raise ValueError(f"cannot cut {height} rows into {world} tiles")
    base, extra = divmod(height, world)
    out, r = [], 0
    for k in range(world):
        n = base + (1 if k < extra else 0)
        out.append((r, r + n))
        r += n
    return out


class TileEngine:
    """Interface a rank's engine offers to `match_tiled` (see include/sgm_mi355x.h, "row tiles")."""

    def begin(self, left, right): raise NotImplementedError
    def new_boundary(self): raise NotImplementedError             # a tensor one hand-over fits in
    def import_boundary(self, forward: bool, buf): raise NotImplementedError
    def sweep(self, forward: bool): raise NotImplementedError
    def export_boundary(self, forward: bool, buf): raise NotImplementedError   # must be complete on return
    def finish(self): raise NotImplementedError                   # -> tensor [rows][W] float32 (this tile's rows)
    def post(self, full): raise NotImplementedError               # full [H][W] -> final [H][W]


class DeviceTileEngine(TileEngine):
    """The product engine: an SGMInstance restricted to the rank's rows; torch owns the device tensors."""

    def __init__(self, device: int, width: int, height: int, option, rows: Tuple[int, int]):
        import torch
        from .sgm import SGMInstance
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.w, self.h, self.rows, self.option = width, height, rows, option
        self.inst = SGMInstance(device)
        if not self.inst.set_rows(*rows):
            raise RuntimeError("sgm_set_rows failed")
        if not self.inst.reset(width, height, option):
            raise RuntimeError("sgm_reset failed")
        self.disp = torch.empty((height, width), dtype=torch.float32, device=self.dev)
        self.nbytes = self.inst.tile_boundary_bytes()

    def _ok(self, ok, what):
        if not ok:
            raise RuntimeError(f"{what} failed")

    def begin(self, left, right):
        """left/right: uint8 [H][W] device tensors (the whole images, replicated on every rank)."""
        self.torch.cuda.current_stream(self.dev).synchronize()     # inputs produced on torch's stream
        self._ok(self.inst.reset(self.w, self.h, self.option), "sgm_reset")       # per frame
        self._left, self._right = left, right                      # keep alive until finish
        self._ok(self.inst.tile_begin(left.data_ptr(), right.data_ptr()), "sgm_tile_begin")

    def new_boundary(self):
        return self.torch.empty(self.nbytes, dtype=self.torch.uint8, device=self.dev)

    def import_boundary(self, forward, buf):
        self._ok(self.inst.tile_import_boundary(forward, buf.data_ptr()), "sgm_tile_import_boundary")

    def sweep(self, forward):
        self._ok(self.inst.tile_sweep(forward), "sgm_tile_sweep")

    def export_boundary(self, forward, buf):
        self._ok(self.inst.tile_export_boundary(forward, buf.data_ptr()), "sgm_tile_export_boundary")
        self._ok(self.inst.synchronize(), "sgm_synchronize")      # the hand-over leaves on another stream

    def finish(self):
        self._ok(self.inst.tile_finish(self.disp.data_ptr()), "sgm_tile_finish")
        self._ok(self.inst.synchronize(), "sgm_synchronize")
        return self.disp[self.rows[0]:self.rows[1]]

    def post(self, full):
        self.torch.cuda.current_stream(self.dev).synchronize()
        self._ok(self.inst.tile_post(full.data_ptr()), "sgm_tile_post")
        self._ok(self.inst.synchronize(), "sgm_synchronize")
        return full

    def close(self):
        self.inst.close()


class _Link:
    """Point-to-point hand-over on one process group.  gloo moves host tensors, so device tensors are staged
    through the host there (1-GPU rehearsals); nccl (= RCCL over xGMI) moves device tensors directly."""

    def __init__(self, dist, group):
        self.dist, self.group = dist, group
        self.host_staged = dist.get_backend(group) == "gloo"
        self.pending = []

    def send(self, buf, dst):
        t = buf.cpu() if (self.host_staged and buf.is_cuda) else buf
        self.pending.append((self.dist.isend(t, dst, group=self.group), t))

    def recv(self, buf, src):
        if self.host_staged and buf.is_cuda:
            t = buf.cpu()
            self.dist.recv(t, src, group=self.group)
            buf.copy_(t)
        else:
            self.dist.recv(buf, src, group=self.group)
        if buf.is_cuda:
            import torch
            torch.cuda.current_stream(buf.device).synchronize()   # the engine imports on its own stream

    def drain(self):
        for req, _keep in self.pending:
            req.wait()
        self.pending.clear()


def make_links(dist):
    """Two process groups over all ranks: one per sweep chain (call on every rank, same order)."""
    world = dist.get_world_size()
    ranks = list(range(world))
    return _Link(dist, dist.new_group(ranks)), _Link(dist, dist.new_group(ranks))


def match_tiled(engine: TileEngine, rank: int, world: int, left, right, height: int, dist=None, links=None):
    """One frame over `world` ranks; returns the final [H][W] disparity map (on every rank).

    `dist` = torch.distributed (initialised) when world > 1; `links` from make_links(dist) (built once, reused
    for every frame)."""
    import torch

    rows = tile_rows(height, world)
    engine.begin(left, right)
    if world > 1 and links is None:
        links = make_links(dist)

    def chain(forward: bool):
        link = links[0 if forward else 1] if world > 1 else None
        prev_rank = rank - 1 if forward else rank + 1            # who hands over to us
        next_rank = rank + 1 if forward else rank - 1            # whom we hand over to
        if 0 <= prev_rank < world:
            buf = engine.new_boundary()
            link.recv(buf, prev_rank)
            engine.import_boundary(forward, buf)
        engine.sweep(forward)
        if 0 <= next_rank < world:
            out = engine.new_boundary()
            engine.export_boundary(forward, out)
            link.send(out, next_rank)

    forward_first = rank < (world + 1) // 2                      # the chain that reaches this rank first
    chain(forward_first)
    chain(not forward_first)

    mine = engine.finish()
    if world == 1:
        full = mine.clone()                                      # not a view of the engine's own buffer
    else:
        for l in links:
            l.drain()
        # rows of all ranks -> one [H][W] map on every rank (tiles differ by at most one row: pad to the largest)
        most = max(r1 - r0 for r0, r1 in rows)
        staged = links[0].host_staged and mine.is_cuda
        src = mine.cpu() if staged else mine
        pad = torch.zeros((most,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=links[0].group)
        full = torch.cat([parts[k][: rows[k][1] - rows[k][0]] for k in range(world)], dim=0)
        if staged:
            full = full.to(mine.device)
    return engine.post(full.contiguous())


def match_tiled_in_process(engines, left, right):
    """The same schedule with all tiles driven by ONE process (engines[k] owns tile k; e.g. N instances on
    one GPU): a rehearsal of the hand-over order without a process group -- the single-GPU parity test of the
    tile mode runs this.  Returns the final [H][W] map."""
    import torch

    n = len(engines)
    for e in engines:
        e.begin(left, right)
    for forward in (True, False):
        order = range(n) if forward else range(n - 1, -1, -1)
        buf = None
        for k in order:
            if buf is not None:
                engines[k].import_boundary(forward, buf)
            engines[k].sweep(forward)
            last = (k == n - 1) if forward else (k == 0)
            if not last:
                buf = engines[k].new_boundary()
                engines[k].export_boundary(forward, buf)
    full = torch.cat([e.finish() for e in engines], dim=0) if n > 1 else engines[0].finish().clone()
    return engines[0].post(full.contiguous())


# ======================================================================================================================
# Frames in flight: the ranks as a systolic pipeline
# ======================================================================================================================
#
# match_tiled above runs ONE frame at a time: while rank 0 sweeps, ranks 1..N-1 wait for its hand-over, so a frame takes
# as long as on one GPU.  With frames in flight the same hand-overs form a pipeline: at step s rank r does
#
#     tile_begin            of frame s                    (census, horizontal paths of its rows, anomalous lines)
#     forward sweep         of frame s - r                (needs the hand-over rank r-1 produced at step s-1)
#     backward sweep        of frame s - (N-1-r)          (needs the hand-over rank r+1 produced at step s-1)
#     -- the exchange of the step (below) is queued here --
#     tile_finish           of frame s - max(r, N-1-r)    (both sweeps of that frame are done: cost sum, WTA, LR check)
#     speckle + median      of frame s - N - 1, on its owner rank only (not on every rank)
#
# and between steps s and s+1 ONE grouped exchange per rank (torch.distributed.batch_isend_irecv: a single RCCL group,
# so the sends and receives of neighbouring ranks cannot deadlock on each other): forward hand-over to r+1, backward
# hand-over to r-1, and the disparity rows of frame s-N -- finished on every rank in an earlier step -- to its owner.
# Every rank issues the same step sequence, so each send meets its receive in the same exchange.  The exchange is queued
# before the step's tile_finish: the next step's sweeps wait for it, and it waits for whatever its slots have queued so
# far, so a cost sum queued in front of it would sit on the chain sweep -> exchange -> sweep that paces the pipeline
# (DESIGN.md section 6).  A frame lives in one of R
# >= N+3 slots per rank (a slot = an SGMInstance restricted to the rank's rows, its stream, its hand-over buffers and a
# [H][W] disparity map); nothing blocks the host: kernels of a slot are ordered by its stream, exchanges run on a
# communication stream, and HIP events order the two (slot -> exchange -> slot).  In the steady state every rank does
# 1/N of every stage of one frame per step; speckle + median of a frame run once, on its owner.

class SlotEngine:
    """What TilePipeline drives: `slots` independent frame contexts on this rank (see DeviceSlotEngine)."""
    slots = 0

    def begin(self, slot, left, right): raise NotImplementedError
    def boundary(self, slot, forward, incoming): raise NotImplementedError   # preallocated hand-over tensor
    def import_boundary(self, slot, forward): raise NotImplementedError      # from boundary(slot, forward, True)
    def sweep(self, slot, forward): raise NotImplementedError
    def export_boundary(self, slot, forward): raise NotImplementedError      # into boundary(slot, forward, False)
    def finish(self, slot): raise NotImplementedError                        # rows of this rank -> frame_map(slot)[r0:r1]
    def frame_map(self, slot): raise NotImplementedError                     # [H][W] float32 tensor of the slot
    def post(self, slot): raise NotImplementedError                          # speckle + median on frame_map(slot), in place
    def exchange(self, dist, ops, slots): raise NotImplementedError          # ops: [("send"|"recv", tensor, peer)]
    def row_views(self, slot, r0, r1): return [self.frame_map(slot)[r0:r1]]  # contiguous tensors holding rows [r0, r1) of the slot's map(s)
    def done(self, slot): return None                                        # event after the slot's last queued work
    def drain(self): pass                                                    # host waits for everything queued


class DeviceSlotEngine(SlotEngine):
    """The product engine: `slots` SGMInstances of this rank's GPU, each restricted to the rank's rows (its planes hold
    only those rows), each on its own HIP stream; a communication stream for the exchanges; HIP events in between."""

    def __init__(self, device: int, width: int, height: int, option, rows: Tuple[int, int], slots: int, host_staged: bool, batch: int = 1):
        import torch
        from .sgm import SGMInstance
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.w, self.h, self.rows, self.option, self.slots, self.batch = width, height, rows, option, slots, batch
        self.host_staged = host_staged              # gloo moves host tensors (rehearsals on a box whose ranks share one GPU)
        self.inst, self.stream, self.maps, self.bufs, self.keep = [], [], [], [], [None] * slots
        # the same guard as sgm_tiles_create: say what does not fit (and what would) before hipMalloc fails half-way through
        from . import tiles as _tiles
        per_slot = _tiles.slot_bytes(rows[0], rows[1], width, height, option, batch)
        free_b, _total = torch.cuda.mem_get_info(self.dev)
        if per_slot and per_slot * slots > free_b:
            fit = free_b // max(1, per_slot // batch) // slots
            raise RuntimeError(f"{slots} slots x batch {batch} of {width}x{height} need about {per_slot * slots / 1e9:.1f} GB on device {device}, "
                               f"{free_b / 1e9:.1f} GB are free: use a batch of at most {fit}, a smaller lead, or more ranks")
        for _ in range(slots):
            i = SGMInstance(device, batch=batch)         # batch > 1: a slot is a batch of frames ([B][H][W] images and maps)
            if not (i.set_rows(*rows) and i.reset(width, height, option)):
                raise RuntimeError("sgm_set_rows / sgm_reset failed")
            self.inst.append(i)
            self.stream.append(torch.cuda.ExternalStream(i.stream, device=self.dev))
            self.maps.append(torch.empty((batch, height, width) if batch > 1 else (height, width), dtype=torch.float32, device=self.dev))
            n = i.tile_boundary_bytes()
            self.bufs.append({(f, inc): torch.empty(n, dtype=torch.uint8, device=self.dev) for f in (True, False) for inc in (True, False)})
        self.comm = torch.cuda.Stream(device=self.dev)

    def _ok(self, ok, what):
        if not ok:
            raise RuntimeError(f"{what} failed")

    def begin(self, slot, left, right):
        t = self.torch
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.dev))                  # the images were produced on torch's stream
        self.stream[slot].wait_event(ev)
        i = self.inst[slot]
        self._ok(i.reset(self.w, self.h, self.option), "sgm_reset")   # per frame; allocates nothing
        self.keep[slot] = (left, right)
        self._ok(i.tile_begin(left.data_ptr(), right.data_ptr()), "sgm_tile_begin")

    def boundary(self, slot, forward, incoming):
        return self.bufs[slot][(forward, incoming)]

    def import_boundary(self, slot, forward):
        self._ok(self.inst[slot].tile_import_boundary(forward, self.bufs[slot][(forward, True)].data_ptr()), "sgm_tile_import_boundary")

    def sweep(self, slot, forward):
        self._ok(self.inst[slot].tile_sweep(forward), "sgm_tile_sweep")

    def export_boundary(self, slot, forward):
        self._ok(self.inst[slot].tile_export_boundary(forward, self.bufs[slot][(forward, False)].data_ptr()), "sgm_tile_export_boundary")

    def finish(self, slot):
        self._ok(self.inst[slot].tile_finish(self.maps[slot].data_ptr()), "sgm_tile_finish")

    def frame_map(self, slot):
        return self.maps[slot]

    def row_views(self, slot, r0, r1):
        m = self.maps[slot]
        return [m[r0:r1]] if self.batch == 1 else [m[b, r0:r1] for b in range(self.batch)]

    def post(self, slot):
        self._ok(self.inst[slot].tile_post(self.maps[slot].data_ptr()), "sgm_tile_post")

    def done(self, slot):
        ev = self.torch.cuda.Event()
        ev.record(self.stream[slot])
        return ev

    def exchange(self, dist, ops, slots):
        """One grouped exchange.  The communication stream first waits for everything queued on the slots whose
        buffers the operations read or overwrite; afterwards those slots' streams wait for the exchange."""
        t = self.torch
        if not ops:
            return
        for s in slots:
            self.comm.wait_event(self.done(s))
        if self.host_staged:
            self.comm.synchronize()
            staged = [(kind, (buf.cpu() if kind == "send" else t.empty(buf.shape, dtype=buf.dtype)), buf, peer) for kind, buf, peer in ops]
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, h, p) for k, h, _, p in staged])
            for r in reqs:
                r.wait()
            with t.cuda.stream(self.comm):
                for kind, host, buf, _ in staged:
                    if kind == "recv":
                        buf.copy_(host)
        else:
            with t.cuda.stream(self.comm):
                reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, b, p) for k, b, p in ops])
                for r in reqs:
                    r.wait()                                        # orders the communication stream, does not block the host
        ev = t.cuda.Event()
        ev.record(self.comm)
        for s in slots:
            self.stream[s].wait_event(ev)

    def drain(self):
        for s in self.stream:
            s.synchronize()
        self.comm.synchronize()

    def close(self):
        self.drain()
        for i in self.inst:
            i.close()


class TilePipeline:
    """The step sequence above for one rank.  run(n_frames, get_frame, on_result):
         get_frame(f)  -> (left, right) of frame f (every rank holds the whole images);
         on_result(f, tensor, event) is called on the OWNER rank of frame f (f mod world) once speckle + median are queued;
         `tensor` ([H][W]) is the slot's map: valid after `event` (None = already complete) and until the slot is reused
         (slots - slots_needed(world, lead) + 1 steps later) -- copy or consume it before."""

    @staticmethod
    def slots_needed(world: int, lead: int = 0) -> int:
        """A frame occupies its slot from tile_begin (step f) to the post pass (step f + lead + world + 1; with one rank there is
        no row gather to wait for and the post pass follows tile_finish in step f + lead)."""
        from . import tiles
        return tiles.slots_needed(world, lead)

    def __init__(self, engine: SlotEngine, rank: int, world: int, height: int, dist=None, lead: int = 0):
        if lead < 0:
            raise ValueError("lead must be >= 0")
        if engine.slots < self.slots_needed(world, lead):
            raise ValueError(f"{world} ranks with a lead of {lead} need at least {self.slots_needed(world, lead)} slots per rank, got {engine.slots}")
        self.e, self.rank, self.world, self.h, self.dist, self.lead = engine, rank, world, height, dist, lead
        self.rows = tile_rows(height, world)

    def run(self, n_frames: int, get_frame, on_result=None, throttle: int = 0):
        """Runs the stream.  The step order is the C library's (sgm_tile_step, csrc/sgm_tile_sched.c -- the same function the
        device pipeline of sgm_tiles.c runs); this method only adapts a SlotEngine to its engine table.

        `lead`: tile_begin of a frame is queued `lead` steps before its first sweep can start.  tile_begin is the longest serial
        chain of a tile (its horizontal lines: W - 1 dependent steps, whatever the number of ranks) and uses a fraction of the
        GPU; queued in the step of the first sweep (lead 0) it is on the critical path of rank 0 and rank N-1 -- every step then
        lasts tile_begin + sweep -- with a lead it runs beside the sweeps of the frames before."""
        from . import tiles
        e, r, N, F, K = self.e, self.rank, self.world, n_frames, self.lead
        R = e.slots

        def begin(slot, frame):
            l, rt = get_frame(frame)
            e.begin(slot, l, rt)

        def exchange(ops, slots):
            # boundary operations first, then the row gather, as the schedule lists them (neighbouring ranks list the
            # operations between them in the same order)
            py = []
            for o in ops:
                kind = "send" if o.kind == tiles.XOP_SEND else "recv"
                if o.buf == tiles.XBUF_BOUNDARY:
                    py.append((kind, e.boundary(o.slot, bool(o.forward), bool(o.incoming)), o.peer))
                else:
                    py += [(kind, v, o.peer) for v in e.row_views(o.slot, o.row_begin, o.row_end)]
            e.exchange(self.dist, py, slots)

        def post(slot, frame):
            e.post(slot)
            if on_result is not None:
                on_result(frame, e.frame_map(slot), e.done(slot))

        eng = tiles.PyEngine(begin, e.import_boundary, e.sweep, e.export_boundary, exchange, e.finish, post)
        step_done = []
        for step in range(tiles.steps_total(F, N, K)):
            if throttle and step >= throttle and step_done[step - throttle] is not None:
                step_done[step - throttle].synchronize()             # bound the host's run-ahead
            eng.step(r, N, self.h, R, K, step, F)
            step_done.append(e.done(step % R) if throttle else None)
        e.drain()


class InProcessGroup:
    """N ranks as N threads of ONE process on one GPU: what a TilePipeline rehearsal passes as `dist` when there is no process
    group.  A send is a device copy into a staging tensor (standing in for the xGMI transfer) queued to the peer with a HIP
    event; the matching receive waits for the event on the receiver's communication stream and copies into its buffer.  Sends
    of a group are queued before its receives are waited for, like a grouped RCCL exchange: no order of the ranks deadlocks.
    `view(rank)` is the object a rank hands to TilePipeline / DeviceSlotEngine.exchange."""

    class _Op:
        def __init__(self, kind, tensor, peer):
            self.kind, self.tensor, self.peer = kind, tensor, peer

    class _View:
        isend, irecv = "send", "recv"

        def __init__(self, group, rank):
            self.group, self.rank = group, rank

        def P2POp(self, kind, tensor, peer):
            return InProcessGroup._Op(kind, tensor, peer)

        def batch_isend_irecv(self, ops):
            import torch
            st = torch.cuda.current_stream() if any(op.tensor.is_cuda for op in ops) else None   # host tensors: the CPU tests
            for op in ops:
                if op.kind == "send":
                    stage, ev = op.tensor.clone(), None
                    if st is not None:
                        ev = torch.cuda.Event()
                        ev.record(st)
                    self.group.q[(self.rank, op.peer)].put((stage, ev))
            for op in ops:
                if op.kind == "recv":
                    stage, ev = self.group.q[(op.peer, self.rank)].get(timeout=self.group.timeout)
                    if ev is not None:
                        st.wait_event(ev)
                    op.tensor.copy_(stage)
                    if ev is not None:
                        stage.record_stream(st)          # allocated on the sender's stream, last read on this one
            return []

    def __init__(self, world: int, timeout: float = 120.0):
        import queue
        self.world, self.timeout = world, timeout
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world) if a != b}

    def view(self, rank: int):
        return InProcessGroup._View(self, rank)


class NullGroup:
    """A `dist` whose exchanges move nothing: ONE rank of an N-rank pipeline runs alone on its GPU, with whatever its hand-over
    buffers hold.  The results are meaningless; the launches, their sizes and their order are exactly the rank's share of the
    work, so the time per step is what that rank's GPU would need if the exchanges cost nothing (bench.py --tile-rank-alone)."""
    isend, irecv = "send", "recv"

    def P2POp(self, kind, tensor, peer):
        return None

    def batch_isend_irecv(self, ops):
        return []
