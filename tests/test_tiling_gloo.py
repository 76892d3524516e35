"""Row-tile schedule (soc_project_stereo_matching_amd/tiling.py) on CPU: the hand-over order, the two chains on
their own process groups and the row gather, over gloo with 2 and 3 ranks.  A toy engine with the same data
dependencies as SGM's vertical/diagonal paths (a recurrence down and up the rows, three shifted 'directions'
per sweep) stands in for the GPU; the tiled result must equal the one-tile result."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


class ToyEngine:
    SHIFTS = (0, 1, -1)

    def __init__(self, h, w, rows):
        import torch
        self.torch, self.h, self.w, self.rows = torch, h, w, rows
        self.log = []

    def begin(self, left, right):
        t = self.torch
        self.left, self.right = left.to(t.int64), right.to(t.int64)
        self.planes = {True: t.zeros((3, self.h, self.w), dtype=t.int64), False: t.zeros((3, self.h, self.w), dtype=t.int64)}
        self.imported = {True: False, False: False}
        self.log.append("begin")

    def new_boundary(self):
        return self.torch.zeros((3, self.w), dtype=self.torch.int64)

    def import_boundary(self, forward, buf):
        r0, r1 = self.rows
        self.planes[forward][:, r0 - 1 if forward else r1] = buf
        self.imported[forward] = True
        self.log.append(("import", forward))

    def sweep(self, forward):
        t = self.torch
        r0, r1 = self.rows
        img = self.left if forward else self.right
        at_edge = (r0 == 0) if forward else (r1 == self.h)
        assert at_edge or self.imported[forward], "sweep before the neighbour's hand-over arrived"
        ys = range(r0, r1) if forward else range(r1 - 1, r0 - 1, -1)
        for k, s in enumerate(self.SHIFTS):
            P = self.planes[forward][k]
            for y in ys:
                prev = y - 1 if forward else y + 1
                if 0 <= prev < self.h:
                    P[y] = img[y] * (k + 1) + t.roll(P[prev], s) % 1000003
                else:
                    P[y] = img[y] * (k + 1)
        self.log.append(("sweep", forward))

    def export_boundary(self, forward, buf):
        r0, r1 = self.rows
        buf.copy_(self.planes[forward][:, r1 - 1 if forward else r0])
        self.log.append(("export", forward))

    def finish(self):
        r0, r1 = self.rows
        return (self.planes[True].sum(0) + self.planes[False].sum(0))[r0:r1].to(self.torch.float32)

    def post(self, full):
        assert tuple(full.shape) == (self.h, self.w)
        return full + 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _images(h, w):
    rng = np.random.default_rng(h * 1000 + w)
    return rng.integers(0, 255, (h, w), dtype=np.uint8), rng.integers(0, 255, (h, w), dtype=np.uint8)


def _worker(rank, world, port, h, w, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from soc_project_stereo_matching_amd.tiling import make_links, match_tiled, tile_rows
    from test_tiling_gloo import ToyEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    links = make_links(dist)
    l, r = _images(h, w)
    eng = ToyEngine(h, w, tile_rows(h, world)[rank])
    for frame in range(2):                                          # links are reused across frames
        full = match_tiled(eng, rank, world, torch.from_numpy(l) + frame, torch.from_numpy(r), h, dist=dist, links=links)
        np.save(f"{out_path}.r{rank}.f{frame}.npy", full.numpy())
    # ranks in the upper half sweep forward first, the others backward first
    first = [e for e in eng.log if isinstance(e, tuple) and e[0] == "sweep"][0][1]
    assert first == (rank < (world + 1) // 2)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_rows_partition():
    from soc_project_stereo_matching_amd.tiling import tile_rows
    for h in (1, 7, 375, 376, 1080):
        for world in (1, 2, 3, 8):
            if h < world:
                with pytest.raises(ValueError):
                    tile_rows(h, world)
                continue
            rows = tile_rows(h, world)
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            sizes = [b - a for a, b in rows]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_tiled_schedule_over_gloo_equals_one_tile(tmp_path, world):
    import torch
    import torch.multiprocessing as mp
    from soc_project_stereo_matching_amd.tiling import match_tiled, match_tiled_in_process, tile_rows
    h, w = 23, 17
    out = str(tmp_path / "t")
    mp.spawn(_worker, args=(world, _free_port(), h, w, out), nprocs=world, join=True)
    l, r = _images(h, w)
    for frame in range(2):
        want = match_tiled(ToyEngine(h, w, (0, h)), 0, 1, torch.from_numpy(l) + frame, torch.from_numpy(r), h).numpy()
        for rank in range(world):                                   # every rank ends with the whole map
            got = np.load(f"{out}.r{rank}.f{frame}.npy")
            assert np.array_equal(got, want), (rank, frame)
        # the in-process rehearsal of the same schedule agrees too
        engines = [ToyEngine(h, w, rows) for rows in tile_rows(h, world)]
        again = match_tiled_in_process(engines, torch.from_numpy(l) + frame, torch.from_numpy(r)).numpy()
        assert np.array_equal(again, want)


# ---------------------------------------------------------------------------------------------------------------------
# Frames in flight: TilePipeline (the systolic schedule) with a toy slot engine over gloo
# ---------------------------------------------------------------------------------------------------------------------

class ToySlotEngine:
    """`slots` ToyEngines behind the SlotEngine interface; the exchange is a plain gloo batch_isend_irecv.  Every call
    is recorded so the test can check the schedule (who posts what, at which step)."""

    def __init__(self, h, w, rows, slots):
        import torch
        self.torch, self.h, self.w, self.rows, self.slots = torch, h, w, rows, slots
        self.eng = [ToyEngine(h, w, rows) for _ in range(slots)]
        self.maps = [torch.zeros((h, w), dtype=torch.float32) for _ in range(slots)]
        self.bufs = [{(f, inc): self.eng[0].new_boundary() for f in (True, False) for inc in (True, False)} for _ in range(slots)]
        self.log = []
        self.exchanges = 0

    def begin(self, slot, left, right):
        self.eng[slot].begin(left, right)
        self.maps[slot].fill_(-7)                                     # stale rows must never survive into a result
        self.log.append(("begin", slot))

    def boundary(self, slot, forward, incoming):
        return self.bufs[slot][(forward, incoming)]

    def import_boundary(self, slot, forward):
        self.eng[slot].import_boundary(forward, self.bufs[slot][(forward, True)])

    def sweep(self, slot, forward):
        self.eng[slot].sweep(forward)

    def export_boundary(self, slot, forward):
        self.eng[slot].export_boundary(forward, self.bufs[slot][(forward, False)])

    def finish(self, slot):
        r0, r1 = self.rows
        self.maps[slot][r0:r1] = self.eng[slot].finish()

    def frame_map(self, slot):
        return self.maps[slot]

    def row_views(self, slot, r0, r1):
        return [self.maps[slot][r0:r1]]

    def post(self, slot):
        assert (self.maps[slot] != -7).all(), "a row of another rank never arrived"
        # owners finish their frames at different speeds: frames then complete out of order across the ranks
        import time
        time.sleep(0.004 * ((slot * 5 + self.rows[0]) % 3))
        self.maps[slot] += 1
        self.log.append(("post", slot))

    def exchange(self, dist, ops, slots):
        if not ops:
            return
        self.exchanges += 1
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, t, p) for k, t, p in ops])
        for r in reqs:
            r.wait()

    def done(self, slot):
        return None

    def drain(self):
        pass


def _pipe_worker(rank, world, port, h, w, n_frames, out_path, lead=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from soc_project_stereo_matching_amd.tiling import TilePipeline, tile_rows
    from test_tiling_gloo import ToySlotEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    l, r = _images(h, w)
    eng = ToySlotEngine(h, w, tile_rows(h, world)[rank], TilePipeline.slots_needed(world, lead))
    pipe = TilePipeline(eng, rank, world, h, dist=dist, lead=lead)
    got = {}
    import time
    for rep in range(2):                                            # the pipeline object is reusable
        pipe.run(n_frames, lambda f: (torch.from_numpy(l) + f, torch.from_numpy(r)),
                 lambda f, t, ev: (got.__setitem__((rep, f), t.clone()),
                                   open(f"{out_path}.order", "a").write(f"{rep} {f} {time.time():.6f}\n")))
    for (rep, f), t in got.items():
        assert f % world == rank                                    # only a frame's owner runs its post pass
        np.save(f"{out_path}.rep{rep}.f{f}.npy", t.numpy())
    assert sum(1 for e in eng.log if e[0] == "post") == 2 * len(range(rank, n_frames, world))
    assert eng.exchanges <= 2 * (n_frames + world + 2 + lead)       # one grouped exchange per step
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,lead", [(2, 5, 0), (3, 7, 0), (3, 1, 0), (4, 2, 0), (8, 11, 0), (3, 7, 2), (2, 1, 3), (4, 9, 1)])
def test_frames_in_flight_pipeline_over_gloo(tmp_path, world, n_frames, lead):
    """The systolic schedule: rank r sweeps frame s-r forward and frame s-(N-1-r) backward at step s, one grouped
    exchange per step, the owner of a frame (f mod N) gathers its rows and runs the post pass.  Every frame -- also
    when there are fewer frames than ranks, and across two runs of the same pipeline -- equals the one-tile result."""
    import torch
    import torch.multiprocessing as mp
    from soc_project_stereo_matching_amd.tiling import match_tiled
    h, w = 23, 17
    out = str(tmp_path / "p")
    mp.spawn(_pipe_worker, args=(world, _free_port(), h, w, n_frames, out, lead), nprocs=world, join=True)
    l, r = _images(h, w)
    for f in range(n_frames):
        want = match_tiled(ToyEngine(h, w, (0, h)), 0, 1, torch.from_numpy(l) + f, torch.from_numpy(r), h).numpy()
        for rep in range(2):
            assert np.array_equal(np.load(f"{out}.rep{rep}.f{f}.npy"), want), (rep, f)
    # every frame of every run was delivered exactly once (the owners' post passes take different times, so the wall-clock
    # order of completion across ranks is not the frame order; results are keyed by frame, not by arrival)
    done = [tuple(int(v) for v in line.split()[:2]) for line in open(f"{out}.order")]
    assert sorted(done) == [(rep, f) for rep in range(2) for f in range(n_frames)]


def test_pipeline_single_rank_and_slot_guard():
    import torch
    from soc_project_stereo_matching_amd.tiling import TilePipeline, match_tiled
    h, w = 19, 11
    l, r = _images(h, w)
    with pytest.raises(ValueError):
        TilePipeline(ToySlotEngine(h, w, (0, h), 1), 0, 1, h)       # one rank needs 2 slots
    with pytest.raises(ValueError):
        TilePipeline(ToySlotEngine(h, w, (0, h), 3), 0, 1, h, lead=2)   # ... + lead
    eng = ToySlotEngine(h, w, (0, h), 2)
    got = {}
    TilePipeline(eng, 0, 1, h).run(4, lambda f: (torch.from_numpy(l) + f, torch.from_numpy(r)),
                                   lambda f, t, ev: got.__setitem__(f, t.clone()))
    assert sorted(got) == [0, 1, 2, 3]
    got2 = {}
    TilePipeline(ToySlotEngine(h, w, (0, h), 4), 0, 1, h, lead=2).run(4, lambda f: (torch.from_numpy(l) + f, torch.from_numpy(r)),
                                                                      lambda f, t, ev: got2.__setitem__(f, t.clone()))
    assert sorted(got2) == [0, 1, 2, 3] and all(torch.equal(got[f], got2[f]) for f in got)
    for f in range(4):
        want = match_tiled(ToyEngine(h, w, (0, h)), 0, 1, torch.from_numpy(l) + f, torch.from_numpy(r), h)
        assert torch.equal(got[f], want)


@pytest.mark.parametrize("world,n_frames,lead", [(2, 5, 0), (3, 7, 2), (5, 3, 1)])
def test_pipeline_ranks_as_threads_of_one_process(world, n_frames, lead):
    """tiling.InProcessGroup: the ranks as threads of one process (what bench.py --tile-ranks-in-process runs on one GPU), the
    exchange as queued copies.  Same results as the one-tile run, whatever order the threads reach their exchanges in."""
    import threading
    import torch
    from soc_project_stereo_matching_amd.tiling import InProcessGroup, TilePipeline, match_tiled, tile_rows
    h, w = 23, 17
    l, r = _images(h, w)
    group = InProcessGroup(world, timeout=60)
    got, errors = {}, []

    def rank_main(rank):
        try:
            eng = ToySlotEngine(h, w, tile_rows(h, world)[rank], TilePipeline.slots_needed(world, lead))
            TilePipeline(eng, rank, world, h, dist=group.view(rank), lead=lead).run(
                n_frames, lambda f: (torch.from_numpy(l) + f, torch.from_numpy(r)), lambda f, t, ev: got.__setitem__(f, (rank, t.clone())))
        except Exception as exc:                                    # noqa: BLE001
            errors.append((rank, repr(exc)))

    th = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    assert sorted(got) == list(range(n_frames))
    for f in range(n_frames):
        want = match_tiled(ToyEngine(h, w, (0, h)), 0, 1, torch.from_numpy(l) + f, torch.from_numpy(r), h)
        assert got[f][0] == f % world and torch.equal(got[f][1], want), f
    assert all(q.empty() for q in group.q.values())                 # every send met its receive
