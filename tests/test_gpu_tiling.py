"""Row-tile mode (one frame over several GPUs, include/sgm_mi355x.h "row tiles") on ONE MI355X: N instances
stand in for N GPUs, the hand-over buffers travel exactly as they would between ranks.  The result must be the
single-GPU result bit for bit -- aggregated cost S on every tile's rows and the final map."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def _engines(w, h, opt, n, honor4=False):
    from soc_project_stereo_matching_amd.tiling import DeviceTileEngine, tile_rows
    out = []
    for rows in tile_rows(h, n):
        e = DeviceTileEngine.__new__(DeviceTileEngine)
        # same as DeviceTileEngine(0, w, h, opt, rows) but with the 4-path switch set before the first reset
        import torch
        from soc_project_stereo_matching_amd.sgm import SGMInstance
        e.torch, e.dev = torch, torch.device("cuda", 0)
        e.w, e.h, e.rows, e.option = w, h, rows, opt
        e.inst = SGMInstance(0)
        e.inst.set_honor_num_paths(honor4)
        assert e.inst.set_rows(*rows)
        assert e.inst.reset(w, h, opt)
        e.disp = torch.empty((h, w), dtype=torch.float32, device=e.dev)
        e.nbytes = e.inst.tile_boundary_bytes()
        out.append(e)
    return out


@pytest.fixture(autouse=True)
def poisoned_census(monkeypatch):
    """A row-tile instance computes the census of its own rows and of the pixels the anomalous lines read, nothing else of the
    replicated images; with this switch the library fills the census buffers with a pattern first, so a kernel that reads a word
    nobody computed shows up as a wrong result instead of passing on stale data."""
    monkeypatch.setenv("SGM_DEBUG_POISON_CENSUS", "1")


@pytest.mark.parametrize("case", [
    # W, H, dmin, dmax, tiles, option overrides
    (130, 47, 2, 50, 2, {}),                       # padded disparity range, odd sizes
    (130, 47, 2, 50, 5, {}),
    (64, 7, 0, 16, 7, {}),                         # one row per tile
    (37, 61, 0, 8, 3, {}),                         # W < H: diagonals wrap several times, planes pre-cleared
    (257, 64, 0, 100, 4, {"is_check_unique": False}),
    (320, 96, 0, 64, 3, {"is_check_lr": False, "is_remove_speckles": False}),
    (450, 375, 0, 64, 8, {}),                      # the reference's own capacity, 8 tiles as on an 8-GPU node
    (200, 40, 0, 256, 2, {}),
], ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}-{c[3]}_n{c[4]}")
def test_tiles_reproduce_the_single_gpu_result(oracle, case):
    import torch
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd.tiling import match_tiled_in_process
    w, h, dmin, dmax, n, over = case
    opt = default_option(dmax, dmin, **over)
    left, right = oracle.synth_pair(w, h, dmax - dmin, 0x71E0 + w + n)
    want = oracle.run(left, right, opt)
    engines = _engines(w, h, opt, n)
    try:
        dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        got = match_tiled_in_process(engines, dl, dr).cpu().numpy()
        for e in engines:                                          # S on the rows each tile owns
            r0, r1 = e.rows
            assert_same(e.inst.read_stage("aggr")[r0:r1], want["aggr"][r0:r1], f"S rows {r0}:{r1}")
        assert_same(got, want["final"], "final")
        # a second frame through the same engines (per-frame Reset inside begin())
        left2, right2 = oracle.synth_pair(w, h, dmax - dmin, 0x71E1 + w + n)
        got2 = match_tiled_in_process(engines, torch.from_numpy(left2).cuda(), torch.from_numpy(right2).cuda())
        assert_same(got2.cpu().numpy(), oracle.run(left2, right2, opt)["final"], "second frame")
    finally:
        for e in engines:
            e.close()


def test_kitti_frame_in_eight_tiles_matches_the_reference_digest(golden_cases):
    """BASELINE.json configs[3] shape: a 1242x375 D=128 frame cut into 8 row tiles (47 / 46 rows) as on an 8-GPU
    node; the final map must hash to the digest the REFERENCE produced for this frame (tests/golden/cases.json)."""
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import sha
    from soc_project_stereo_matching_amd.tiling import match_tiled_in_process
    case = golden_cases["c2_kitti_1242x375_d128"]
    w, h, d = case["w"], case["h"], case["d"]
    left, right = S.synth_pair(w, h, d, case["seed"])
    engines = _engines(w, h, S.default_option(d), 8)
    try:
        got = match_tiled_in_process(engines, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda())
        assert sha(got.cpu().numpy()) == case["sha256"]["final"]
    finally:
        for e in engines:
            e.close()


def test_tiles_four_path_mode(oracle):
    import torch
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd.tiling import match_tiled_in_process
    w, h, d = 160, 50, 32
    opt = default_option(d, num_paths=4)
    left, right = oracle.synth_pair(w, h, d, 0x4A7)
    oracle.set_honor_num_paths(True)
    engines = _engines(w, h, opt, 3, honor4=True)
    try:
        want = oracle.run(left, right, opt)["final"]
        assert engines[0].inst.tile_boundary_bytes() == w * 32      # one direction per sweep, Dp = 32
        got = match_tiled_in_process(engines, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda())
        assert_same(got.cpu().numpy(), want, "4-path tiled")
    finally:
        oracle.set_honor_num_paths(False)
        for e in engines:
            e.close()


def test_tile_mode_guards():
    import soc_project_stereo_matching_amd as S
    i = S.SGMInstance(0)
    try:
        opt = S.default_option(16)
        assert i.set_rows(10, 20)
        assert not i.reset(64, 15, opt)                            # tile beyond the frame
        assert i.reset(64, 32, opt)
        assert i.match(np.zeros((32, 64), np.uint8), np.zeros((32, 64), np.uint8)) is None   # whole-frame call refused
        assert not i.tile_sweep(True)                              # no tile_begin yet
        assert i.set_rows(0, 0) and i.reset(64, 32, opt)
        assert i.match(np.zeros((32, 64), np.uint8), np.zeros((32, 64), np.uint8)) is not None
        assert not i.set_rows(5, 3)
    finally:
        i.close()


def test_slot_engines_refuse_what_does_not_fit_before_allocating():
    """Round 3's 2-rank rehearsal died inside hipMalloc on 34 GB of planes for one of seven slots.  Both hosts of the tile pipeline now
    check sgm_tile_slot_bytes x slots against the free device memory first and say which batch would fit: the Python engine
    (tiling.DeviceSlotEngine) raises, the C pipeline (sgm_tiles_create) returns NULL with the same message."""
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd import tiles
    from soc_project_stereo_matching_amd.tiling import DeviceSlotEngine, TilePipeline, tile_rows
    w, h, d, world, batch = 3840, 2160, 128, 2, 16                  # 7 slots x 16 frames x 4.8 GB: over half a terabyte
    opt = S.default_option(d)
    slots = TilePipeline.slots_needed(world, 2)
    need = tiles.slot_bytes(*tile_rows(h, world)[0], w, h, opt, batch) * slots
    assert need > 500e9
    with pytest.raises(RuntimeError, match=r"need about 5\d\d\.\d GB .* use a batch of at most \d+"):
        DeviceSlotEngine(0, w, h, opt, tile_rows(h, world)[0], slots, host_staged=False, batch=batch)
    with pytest.raises(RuntimeError, match="sgm_tiles_create failed"):
        tiles.TilesPipeline(0, 0, world, w, h, opt, batch=batch, lead=2, transport=tiles.NullTransport().struct)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, w, h, d, seed, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd.tiling import DeviceTileEngine, make_links, match_tiled, tile_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    links = make_links(dist)
    opt = S.default_option(d)
    eng = DeviceTileEngine(0, w, h, opt, tile_rows(h, world)[rank])      # every rank on the box's one GPU
    for k in range(2):
        left, right = S.synth_pair(w, h, d, seed + k)
        full = match_tiled(eng, rank, world, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), h,
                           dist=dist, links=links)
        if rank == world - 1:
            np.save(out_path + f".{k}.npy", full.cpu().numpy())
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_three_ranks_over_a_process_group(tmp_path, oracle):
    """The distributed wrapper end to end: three processes (sharing this box's one GPU), hand-overs and the row
    gather over a gloo group (host-staged); on a multi-GPU node the same code runs with backend nccl (= RCCL)."""
    import torch.multiprocessing as mp
    from oracle.pyoracle import default_option
    w, h, d, seed, world = 300, 70, 48, 0xBEEF, 3
    out = str(tmp_path / "tiled")
    mp.spawn(_rank_main, args=(world, _free_port(), w, h, d, seed, out), nprocs=world, join=True)
    for k in range(2):
        l, r = oracle.synth_pair(w, h, d, seed + k)
        assert_same(np.load(out + f".{k}.npy"), oracle.run(l, r, default_option(d))["final"], f"frame {k}")


def test_match_tiled_with_one_rank_returns_its_own_map(oracle):
    """world == 1 through match_tiled proper (the distributed wrapper, not the in-process rehearsal): the result must
    not alias the engine's buffer (round 1 returned a view that the next frame overwrote)."""
    import torch
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd.tiling import DeviceTileEngine, match_tiled
    w, h, d = 260, 90, 48
    opt = default_option(d)
    eng = DeviceTileEngine(0, w, h, opt, (0, h))
    try:
        outs, wants = [], []
        for k in range(2):
            l, r = oracle.synth_pair(w, h, d, 0x1AB0 + k)
            wants.append(oracle.run(l, r, opt)["final"])
            outs.append(match_tiled(eng, 0, 1, torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), h))
        for k in range(2):
            assert_same(outs[k].cpu().numpy(), wants[k], f"frame {k}")
    finally:
        eng.close()


@pytest.mark.parametrize("case", [(300, 70, 48, 1, 5), (1242, 375, 128, 1, 4), (37, 61, 8, 1, 3)],
                         ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}")
def test_slot_pipeline_on_one_rank(oracle, case):
    """TilePipeline + DeviceSlotEngine with one rank: slots on their own streams, events instead of host syncs, result
    delivered through on_result with an event.  Every frame against the oracle."""
    import torch
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd.tiling import DeviceSlotEngine, TilePipeline
    w, h, d, world, n = case
    opt = default_option(d, min_speckle_area=20)
    frames = [oracle.synth_pair(w, h, d, 0x51D0 + k) for k in range(n)]
    dev = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in frames]
    eng = DeviceSlotEngine(0, w, h, opt, (0, h), TilePipeline.slots_needed(world), host_staged=False)
    got = {}

    def on_result(f, t, ev):
        ev.synchronize()
        got[f] = t.cpu().numpy().copy()

    try:
        TilePipeline(eng, 0, 1, h).run(n, lambda f: dev[f], on_result)
        for k in range(n):
            assert_same(got[k], oracle.run(frames[k][0], frames[k][1], opt)["final"], f"frame {k}")
    finally:
        eng.close()


def _pipe_rank_main(rank, world, port, w, h, d, seed, n_frames, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd.tiling import DeviceSlotEngine, TilePipeline, tile_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt = S.default_option(d)
    eng = DeviceSlotEngine(0, w, h, opt, tile_rows(h, world)[rank], TilePipeline.slots_needed(world), host_staged=True)   # every rank on the box's one GPU
    pairs = [S.synth_pair(w, h, d, seed + k) for k in range(n_frames)]
    dev = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in pairs]

    def on_result(f, t, ev):
        ev.synchronize()
        np.save(out_path + f".{f}.npy", t.cpu().numpy())

    TilePipeline(eng, rank, world, h, dist=dist).run(n_frames, lambda f: dev[f], on_result)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_three_rank_pipeline_with_frames_in_flight(tmp_path, oracle):
    """The systolic schedule end to end with the device engine: three processes (sharing this box's one GPU), five
    frames in flight, grouped exchanges over a gloo group (host-staged); on a multi-GPU node the same code runs with
    backend nccl (= RCCL).  Frames complete on their owner ranks (frame f on rank f mod 3)."""
    import torch.multiprocessing as mp
    from oracle.pyoracle import default_option
    w, h, d, seed, world, n = 300, 70, 48, 0xF1F0, 3, 5
    out = str(tmp_path / "pipe")
    mp.spawn(_pipe_rank_main, args=(world, _free_port(), w, h, d, seed, n, out), nprocs=world, join=True)
    for k in range(n):
        l, r = oracle.synth_pair(w, h, d, seed + k)
        assert_same(np.load(out + f".{k}.npy"), oracle.run(l, r, default_option(d))["final"], f"frame {k}")


def test_rccl_grouped_exchange_on_one_gpu():
    """The RCCL code path of DeviceSlotEngine.exchange -- P2POp list, ONE grouped batch_isend_irecv, work.wait() under the
    communication stream, HIP events against the slots' own streams -- with the only RCCL point-to-point a single GPU can
    do: a world-size-1 nccl group whose rank sends to and receives from itself (tools/nccl_self_exchange.py, in a process of
    its own so that the process group does not leak into the test session).  Not the xGMI transport."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_self_exchange.py")], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "nccl self-exchange ok" in p.stdout


@pytest.mark.parametrize("case", [
    # w, h, d, ranks, batch, steps, lead, extras
    (300, 70, 48, 3, 2, 4, 0, {}),
    (211, 96, 64, 2, 3, 3, 2, {}),
    (90, 131, 24, 4, 2, 5, 1, {}),                 # W < H: ghost rows / plane memset per frame of the batch
    (300, 70, 48, 3, 2, 3, 0, {"right": True}),    # right reference view: per-frame row copies of the mirrored LR check
    (160, 50, 40, 2, 4, 2, 0, {"paths4": True}),   # one direction per sweep in the hand-over
], ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}_r{c[3]}_b{c[4]}_l{c[6]}" + "".join("_" + k for k in c[7]))
def test_pipeline_of_batched_tiles_ranks_as_threads(oracle, case):
    """Row tiles of BATCHES of frames (sgm_set_batch on a row-tile instance: one launch per stage for the same tile of B frames;
    hand-over buffers and row gathers carry B frames), the ranks as threads of this process on the one GPU
    (tiling.InProcessGroup), frames in flight, tile_begin queued `lead` steps ahead.  Every frame against the oracle."""
    import threading
    import torch
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd.tiling import DeviceSlotEngine, InProcessGroup, TilePipeline, tile_rows
    w, h, d, ranks, B, steps, lead, extra = case
    opt = default_option(d, min_speckle_area=20, num_paths=4 if extra.get("paths4") else 8)
    frames = [[oracle.synth_pair(w, h, d, 0x7B00 + 16 * k + j) for j in range(B)] for k in range(steps)]
    dev = [(torch.from_numpy(np.stack([p[0] for p in fr])).cuda(), torch.from_numpy(np.stack([p[1] for p in fr])).cuda()) for fr in frames]
    torch.cuda.synchronize()
    group = InProcessGroup(ranks, timeout=60)
    got, errors = {}, []

    def rank_main(r):
        eng = None
        try:
            torch.cuda.set_device(0)
            eng = DeviceSlotEngine(0, w, h, opt, tile_rows(h, ranks)[r], TilePipeline.slots_needed(ranks, lead), host_staged=False, batch=B)
            for i in eng.inst:
                i.set_honor_num_paths(bool(extra.get("paths4")))
                i.set_reference_view(bool(extra.get("right")))
                assert i.reset(w, h, opt)

            def on_result(f, t, ev):
                ev.synchronize()
                got[f] = t.cpu().numpy().copy()

            TilePipeline(eng, r, ranks, h, dist=group.view(r), lead=lead).run(steps, lambda f: dev[f], on_result)
        except Exception as exc:                                    # noqa: BLE001
            errors.append((r, repr(exc)))
        finally:
            if eng is not None:
                eng.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    assert not errors, errors
    assert sorted(got) == list(range(steps))
    try:
        oracle.set_honor_num_paths(bool(extra.get("paths4")))
        oracle.set_reference_view(bool(extra.get("right")))
        for k in range(steps):
            for j in range(B):
                assert_same(got[k][j], oracle.run(frames[k][j][0], frames[k][j][1], opt)["final"], f"step {k} frame {j}")
    finally:
        oracle.set_honor_num_paths(False)
        oracle.set_reference_view(False)


# ======================================================================================================================
# the C host of the pipeline (include/sgm_tiles.h): sgm_tiles_create / _submit / _finish, local and RCCL transports
# ======================================================================================================================

def _c_ranks(oracle, w, h, d, world, batch, lead, n_steps, opt=None, in_flight=0):
    """`world` C pipelines as threads on the one GPU (world == 1: no transport); returns ({step: [batch][H][W] map}, wanted)."""
    import threading
    import torch
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd import tiles
    opt = opt or default_option(d)
    pairs = [[oracle.synth_pair(w, h, d, 0xC711E + 97 * k + j) for j in range(batch)] for k in range(n_steps)]
    dev = [(torch.from_numpy(np.stack([p[0] for p in ps])).cuda(), torch.from_numpy(np.stack([p[1] for p in ps])).cuda()) for ps in pairs]
    torch.cuda.synchronize()
    ring_frames = (n_steps + world - 1) // world
    rings = [torch.full((ring_frames, batch, h, w), -7.0, dtype=torch.float32, device="cuda") for _ in range(world)]
    group = tiles.LocalGroup(world, 0) if world > 1 else None
    errors = []

    def rank_main(r):
        try:
            tr = group.transport(r) if group else None
            need = tiles.slots_needed(world, lead)
            pipe = tiles.TilesPipeline(0, r, world, w, h, opt, batch=batch, lead=lead, spare=max(1, in_flight - need + 1),
                                       throttle=need, transport=tr)
            assert pipe.info()[:2] == tiles.tile_rows(h, world, r)
            pipe.result_ring(rings[r].data_ptr(), ring_frames)
            for k in range(n_steps):
                assert pipe.submit(dev[k][0].data_ptr(), dev[k][1].data_ptr()), f"submit {k}"
            assert pipe.finish()
            pipe.close()
            if tr is not None:
                tr.close()
        except BaseException as e:                                    # noqa: BLE001
            errors.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not any(t.is_alive() for t in th), "a rank is stuck"
    if group:
        group.close()
    assert not errors, errors
    got = {k: rings[k % world][(k // world) % ring_frames].cpu().numpy() for k in range(n_steps)}
    want = {k: [oracle.run(l, r, opt)["final"] for l, r in pairs[k]] for k in range(n_steps)}
    return got, want


@pytest.mark.parametrize("case", [
    # W, H, D, ranks, batch, lead, frames
    (300, 70, 48, 1, 1, 2, 5),
    (300, 70, 48, 2, 1, 0, 6),
    (160, 50, 32, 3, 2, 2, 7),
    (37, 61, 8, 4, 1, 1, 9),                       # W < H
    (130, 47, 48, 5, 1, 3, 11),
], ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}_n{c[3]}_b{c[4]}_lead{c[5]}")
def test_c_pipeline_ranks_as_threads(oracle, case):
    """The library's own pipeline, N ranks as threads connected by its local transport: every frame of the stream -- more frames
    than slots, so slots and hand-over buffers are reused -- must be the single-GPU result bit for bit."""
    w, h, d, world, batch, lead, n = case
    got, want = _c_ranks(oracle, w, h, d, world, batch, lead, n)
    for k in range(n):
        for b in range(batch):
            assert_same(got[k][b], want[k][b], f"frame {k}.{b}")


def test_c_pipeline_kitti_in_eight_tiles_matches_the_reference_digest():
    """BASELINE.json configs[3]: KITTI frames in 8 row tiles through the C pipeline (8 threads standing in for 8 GPUs); the maps
    must hash to the digests the REFERENCE produced (tests/golden/bench_frames.json holds them for the bench's seeds)."""
    import json
    import threading
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import sha
    from soc_project_stereo_matching_amd import tiles
    with open(os.path.join(ROOT, "tests", "golden", "bench_frames.json")) as f:
        wl = json.load(f)["workloads"]["kitti_1242x375_d128_p8"]
    w, h, d, seed0 = 1242, 375, 128, 0x5EED0002
    world, n_steps, lead = 8, 10, 2
    opt = S.default_option(d)
    dev = []
    for k in range(n_steps):
        l, r = S.synth_pair(w, h, d, seed0 + k)
        dev.append((torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()))
    torch.cuda.synchronize()
    ring_frames = 2
    rings = [torch.zeros((ring_frames, 1, h, w), dtype=torch.float32, device="cuda") for _ in range(world)]
    group = tiles.LocalGroup(world, 0)
    errors = []

    def rank_main(r):
        try:
            tr = group.transport(r)
            pipe = tiles.TilesPipeline(0, r, world, w, h, opt, batch=1, lead=lead, throttle=4, transport=tr)
            pipe.result_ring(rings[r].data_ptr(), ring_frames)
            for k in range(n_steps):
                assert pipe.submit(dev[k][0].data_ptr(), dev[k][1].data_ptr())
            assert pipe.finish()
            pipe.close()
            tr.close()
        except BaseException as e:                                    # noqa: BLE001
            errors.append((r, repr(e)))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    group.close()
    assert not errors, errors
    for k in range(n_steps):
        assert sha(rings[k % world][(k // world) % ring_frames][0].cpu().numpy()) == wl["frames"][str(seed0 + k)]["sha256"]["final"], k


def test_c_rccl_transport_on_one_gpu():
    """The product transport of the C host with the REAL librccl, in the only form one GPU allows: a communicator of one rank
    whose grouped sends meet its own receives (dlopen / ncclCommInitRank / ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on
    a stream of the library).  Two ranks on two GPUs are the driver's multi-GPU run."""
    import torch
    from soc_project_stereo_matching_amd import tiles
    uid = tiles.rccl_unique_id()
    assert len(uid) == tiles.ID_BYTES and not uid.startswith(b"stub-rccl")
    tr = tiles.rccl_transport(uid, 0, 1, 0)
    st = torch.cuda.Stream()
    a = torch.arange(1 << 20, dtype=torch.uint8, device="cuda")
    b = torch.arange(4096, dtype=torch.float32, device="cuda")
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    st.wait_stream(torch.cuda.current_stream())
    sp = st.cuda_stream
    for _ in range(3):                                                # the same communicator, several groups
        ra.zero_(); rb.zero_()
        st.wait_stream(torch.cuda.current_stream())
        assert tr.group_start(tr.ctx) == 0
        assert tr.send(tr.ctx, a.data_ptr(), a.numel(), 0, sp) == 0 and tr.recv(tr.ctx, ra.data_ptr(), ra.numel(), 0, sp) == 0
        assert tr.send(tr.ctx, b.data_ptr(), b.numel() * 4, 0, sp) == 0 and tr.recv(tr.ctx, rb.data_ptr(), rb.numel() * 4, 0, sp) == 0
        assert tr.group_end(tr.ctx) == 0
        st.synchronize()
        assert torch.equal(a, ra) and torch.equal(b, rb)
    tr.close()
