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
