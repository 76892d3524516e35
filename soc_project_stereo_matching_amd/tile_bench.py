"""bench.py --mode tiles: every frame cut into WORLD_SIZE row tiles, one per GPU, frames in flight through the ranks.

The pipeline is the C library's (include/sgm_tiles.h: sgm_tiles_create / _submit / _finish -- the step schedule, the slots, their
streams and events, the hand-over buffers and the RCCL transport are all C; this module submits frames and reads results).
Boundary hand-overs + row gather are one grouped RCCL exchange per step; speckle + median run on the frame's owner rank.  A
step is one frame; `value` = W*H*D*8*frames / elapsed (strong scaling: the same frames whatever the number of GPUs).
torch.distributed carries the RCCL id to the ranks, the barriers and the timing all-reduce -- nothing on the data path.
--tile-host python (or SGM_BENCH_BACKEND=gloo, which needs it) runs the same schedule with tiling.TilePipeline's Python engine
instead: gloo stages the hand-overs through the host, which is how the multi-PROCESS path is rehearsed on a box whose ranks
share one GPU (RCCL refuses two ranks on one device).

Verification: the last frames every rank owns are snapshotted on the device inside the timed region and hashed afterwards
against the digests the reference's own C produced for their seeds (tests/golden/bench_frames.json)."""
import hashlib
import json
import os
import time

import numpy as np

PATHS = 8
HBM_PEAK_GBS = 8000.0


def _inputs(S, torch, w, h, d, seed, digests, B):
    """Device-resident synthetic input of the pipeline: a few distinct batches of B pairs ([B][H][W], or [H][W] for B = 1) whose
    seeds have reference digests; returns (get_frame(f), seeds_of(f))."""
    n_known = len(digests) if digests else 0
    n_batches = max(1, min(4, n_known // B)) if n_known >= B else 2
    batches, seeds = [], []
    for k in range(n_batches):
        ps = [S.synth_pair(w, h, d, seed + k * B + j) for j in range(B)]
        seeds.append([seed + k * B + j for j in range(B)])
        if B == 1:
            batches.append((torch.from_numpy(ps[0][0]).cuda(), torch.from_numpy(ps[0][1]).cuda()))
        else:
            batches.append((torch.from_numpy(np.stack([p[0] for p in ps])).cuda(), torch.from_numpy(np.stack([p[1] for p in ps])).cuda()))
    torch.cuda.synchronize()
    return (lambda f: batches[f % n_batches]), (lambda f: seeds[f % n_batches])


def _verify(snaps, seeds_of, digests, B):
    digests = digests or {}
    n_ok = n_bad = n_unpinned = 0
    for f, t in sorted(snaps.items()):
        maps = t.cpu().numpy().reshape((B,) + tuple(t.shape[-2:]))
        for j, sd in enumerate(seeds_of(f)):
            if sd not in digests:
                n_unpinned += 1
            elif hashlib.sha256(maps[j].tobytes()).hexdigest() == digests[sd]:
                n_ok += 1
            else:
                n_bad += 1
    return n_ok, n_bad, n_unpinned


def run_tiles(args, init_dist, WORKLOADS, golden_digests):
    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S
    from .tiling import DeviceSlotEngine, TilePipeline, tile_rows

    world, rank, local_rank, backend = init_dist(args)
    if world > 1:
        # every rank takes part in one collective before the first (partial) point-to-point exchange of the pipeline
        warm = torch.zeros(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(warm)
    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    lead, B = args.tile_lead, max(1, args.batch or 1)
    slots = max(TilePipeline.slots_needed(world, lead), args.in_flight or 0)
    eng = DeviceSlotEngine(local_rank, w, h, opt, tile_rows(h, world)[rank], slots, host_staged=(world > 1 and backend != "nccl"), batch=B)
    pipe = TilePipeline(eng, rank, world, h, dist=dist if world > 1 else None, lead=lead)

    digests = golden_digests(args.workload)
    get, seeds_of = _inputs(S, torch, w, h, d, seed, digests, B)

    def barrier():
        if world > 1:
            dist.barrier()

    keep_last = 2 * world                                            # frames snapshotted for verification
    snaps = {}

    def on_result(f, tensor, event):
        if f >= args.steps - keep_last:
            with torch.cuda.stream(eng.stream[f % slots]):
                snaps[f] = tensor.clone()

    pipe.run(max(args.warmup, 1), get)                               # untimed: first-use allocations, RCCL connections
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(args.steps, get, on_result, throttle=slots)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    n_ok, n_bad, n_unpinned = _verify(snaps, seeds_of, digests, B)
    if world > 1:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, float(n_ok), float(n_bad), float(n_unpinned)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        n_ok, n_bad, n_unpinned = int(t[1].item()), int(t[2].item()), int(t[3].item())
    eng.close()

    if rank == 0:
        print(json.dumps(_line(args, B, w, h, d, world, tile_rows, slots, elapsed, n_ok, n_bad, n_unpinned, keep_last,
                               f"{world} row tiles per frame; per step one grouped exchange per rank over {backend}: boundary path "
                               "costs to both neighbours + finished rows to the frame's owner; speckle+median on the owner", world)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _line(args, B, w, h, d, world, tile_rows, slots, elapsed, n_ok, n_bad, n_unpinned, keep_last, sharding, n_gpus):
    frames = args.steps * B                                          # a step = one slot = B frames
    cells = w * h * d
    dp = -(-d // 16) * 16
    value = cells * PATHS * frames / elapsed / 1e6
    moved = w * h * dp * 16                                      # planes written once + read once
    line = {
        "metric": "Mdisp/s (W*H*D*paths per second), fps beside it",
        "value": round(value, 1), "unit": "Mdisp/s", "fps": round(frames / elapsed, 2),
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8/u16 integer min-plus (f32 sub-pixel tail)", "data": "synthetic",
        "config": {"workload": args.workload, "mode": "tiles", "width": w, "height": h, "disparity_range": d, "paths": PATHS,
                   "stages": "census+cost+aggregate8+wta+lrcheck+speckle+median", "frames": frames, "frames_per_step": B,
                   "tile_rows": tile_rows(h, world), "slots_per_rank": slots, "begin_lead_steps": args.tile_lead,
                   "sharding": sharding},
        "roofline": {"bound": "hbm", "kernel": "whole frame (all kernels, all ranks)", "achieved": round(moved * frames / elapsed / 1e9, 1),
                     "peak": HBM_PEAK_GBS * n_gpus, "unit": "GB/s", "frac": round(moved * frames / elapsed / 1e9 / (HBM_PEAK_GBS * n_gpus), 4),
                     "traffic": None, "algorithmic_bytes": "W*H*Dp*16 per frame (eight u8 planes written once and read once)"},
        "frames_verified": n_ok, "frames_mismatched": n_bad, "frames_without_reference_digest": n_unpinned,
        "verified_against_golden": (n_bad == 0 and n_ok > 0 and n_unpinned == 0),
        "verification": f"sha256 of the frames of the last {keep_last} steps of the timed region (snapshotted on their owner ranks) vs the "
                        "reference's own C, tests/golden/bench_frames.json",
        "cpu_baseline": None,
    }
    return line


def run_tiles_in_process(args, ranks, WORKLOADS, golden_digests):
    """The same pipeline with `ranks` tile ranks as THREADS of this process on ONE GPU (tiling.InProcessGroup: device copies
    stand in for the xGMI transfers).  Not a multi-GPU measurement -- it shows what the schedule itself costs: `ranks` tiles per
    frame, every hand-over and the row gather in place, against the one-rank pipeline on the same GPU."""
    import threading
    import torch
    import soc_project_stereo_matching_amd as S
    from .tiling import DeviceSlotEngine, InProcessGroup, TilePipeline, tile_rows

    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    lead, B = args.tile_lead, max(1, args.batch or 1)
    slots = max(TilePipeline.slots_needed(ranks, lead), args.in_flight or 0)
    group = InProcessGroup(ranks)
    digests = golden_digests(args.workload)
    get, seeds_of = _inputs(S, torch, w, h, d, seed, digests, B)
    keep_last = 2 * ranks
    snaps, errors, times = {}, [], [0.0, 0.0]
    bar = threading.Barrier(ranks)

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            eng = DeviceSlotEngine(0, w, h, opt, tile_rows(h, ranks)[r], slots, host_staged=False, batch=B)
            pipe = TilePipeline(eng, r, ranks, h, dist=group.view(r), lead=lead)

            def on_result(f, tensor, event):
                if f >= args.steps - keep_last:
                    with torch.cuda.stream(eng.stream[f % slots]):
                        snaps[f] = tensor.clone()

            pipe.run(max(args.warmup, 1), get)
            torch.cuda.synchronize()
            bar.wait()
            if r == 0:
                times[0] = time.perf_counter()
            pipe.run(args.steps, get, on_result, throttle=slots)
            torch.cuda.synchronize()
            bar.wait()
            if r == 0:
                times[1] = time.perf_counter()
            eng.close()
        except Exception as exc:                                      # noqa: BLE001
            errors.append((r, repr(exc)))
            bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise RuntimeError(f"in-process tile ranks failed: {errors}")
    elapsed = times[1] - times[0]
    n_ok, n_bad, n_unpinned = _verify(snaps, seeds_of, digests, B)
    line = _line(args, B, w, h, d, ranks, tile_rows, slots, elapsed, n_ok, n_bad, n_unpinned, keep_last,
                 f"{ranks} row tiles per frame driven by {ranks} threads of one process on ONE GPU (device copies stand in for xGMI): "
                 "the schedule's own cost, not a multi-GPU measurement", 1)
    line["config"]["mode"] = f"tiles, {ranks} in-process ranks on one GPU"
    print(json.dumps(line), flush=True)


def run_tile_rank_alone(args, spec, WORKLOADS):
    """`spec` = "r/N": rank r of an N-rank pipeline alone on this GPU, exchanges replaced by nothing (tiling.NullGroup).  Prints the
    rank's time per frame: 1 / max over the ranks of it is the rate N GPUs could reach with the exchanges fully hidden -- a
    projection from measured single-GPU work, NOT a multi-GPU measurement (results are not verified: there are none)."""
    import torch
    import soc_project_stereo_matching_amd as S
    from .tiling import DeviceSlotEngine, NullGroup, TilePipeline, tile_rows
    r, n = (int(v) for v in spec.split("/"))
    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    lead, B = args.tile_lead, max(1, args.batch or 1)
    slots = max(TilePipeline.slots_needed(n, lead), args.in_flight or 0)
    eng = DeviceSlotEngine(0, w, h, opt, tile_rows(h, n)[r], slots, host_staged=False, batch=B)
    pipe = TilePipeline(eng, r, n, h, dist=NullGroup(), lead=lead)
    get, _ = _inputs(S, torch, w, h, d, seed, None, B)
    pipe.run(max(args.warmup, 1), get)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(args.steps, get, None, throttle=slots)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng.close()
    print(json.dumps({"tile_rank_alone": spec, "lead": lead, "slots": slots, "workload": args.workload, "rows": tile_rows(h, n)[r], "frames": args.steps * B,
                      "frames_per_step": B, "ms_per_frame": round(el / (args.steps * B) * 1e3, 4),
                      "fps_if_this_rank_were_the_slowest": round(args.steps * B / el, 2),
                      "note": "one rank's share of an N-rank pipeline alone on one GPU, exchanges skipped: a projection input, not a result"}),
          flush=True)


# ======================================================================================================================
# the C host (include/sgm_tiles.h)
# ======================================================================================================================

def _c_pipeline(S, tiles, device, rank, world, w, h, opt, B, lead, in_flight, transport):
    need = tiles.slots_needed(world, lead)
    spare = max(1, (in_flight or 0) - need + 1)
    slots = need + spare - 1
    pipe = tiles.TilesPipeline(device, rank, world, w, h, opt, batch=B, lead=lead, spare=spare, throttle=slots, transport=transport)
    return pipe, slots


def _c_run(torch, pipe, get, n_steps):
    for f in range(n_steps):
        l, r = get(f)
        if not pipe.submit(l.data_ptr(), r.data_ptr()):
            raise RuntimeError(f"sgm_tiles_submit failed at frame {f}")
    if not pipe.finish():                                            # queues the remaining steps and waits for every stream
        raise RuntimeError("sgm_tiles_finish failed")


def _c_snaps(ring, ring_frames, rank, world, n_steps, keep_last):
    """The frames of the last `keep_last` steps this rank owns, as the result ring holds them after the stream has ended."""
    mine = [f for f in range(max(0, n_steps - keep_last), n_steps) if f % world == rank][-ring_frames:]
    return {f: ring[(f // world) % ring_frames] for f in mine}


def run_tiles_c(args, init_dist, WORKLOADS, golden_digests):
    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S
    from . import tiles
    from .tiling import tile_rows

    world, rank, local_rank, backend = init_dist(args)
    if world > 1 and backend != "nccl":
        raise SystemExit("the C tile host moves device buffers over RCCL: use the nccl backend, or --tile-host python for a gloo rehearsal")
    transport = None
    if world > 1:
        uid = [tiles.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)                       # 128 bytes, once: the only thing torch.distributed moves for the pipeline
        transport = tiles.rccl_transport(uid[0], rank, world, local_rank)
    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    lead, B = args.tile_lead, max(1, args.batch or 1)
    pipe, slots = _c_pipeline(S, tiles, local_rank, rank, world, w, h, opt, B, lead, args.in_flight, transport)
    digests = golden_digests(args.workload)
    get, seeds_of = _inputs(S, torch, w, h, d, seed, digests, B)
    keep_last, ring_frames = 2 * world, 2
    ring = torch.zeros((ring_frames, B, h, w), dtype=torch.float32, device="cuda")
    pipe.result_ring(ring.data_ptr(), ring_frames)

    def barrier():
        if world > 1:
            dist.barrier()

    _c_run(torch, pipe, get, max(args.warmup, 1))                    # untimed: first-use allocations, RCCL connections
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _c_run(torch, pipe, get, args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    n_ok, n_bad, n_unpinned = _verify(_c_snaps(ring, ring_frames, rank, world, args.steps, keep_last), seeds_of, digests, B)
    if world > 1:
        t = torch.tensor([elapsed, float(n_ok), float(n_bad), float(n_unpinned)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        n_ok, n_bad, n_unpinned = int(t[1].item()), int(t[2].item()), int(t[3].item())
    pipe.close()
    if transport is not None:
        transport.close()
    if rank == 0:
        line = _line(args, B, w, h, d, world, tile_rows, slots, elapsed, n_ok, n_bad, n_unpinned, keep_last,
                     f"{world} row tiles per frame; per step one grouped RCCL exchange per rank (ncclSend / ncclRecv from the C host): "
                     "boundary path costs to both neighbours + finished rows to the frame's owner; speckle+median on the owner", world)
        line["config"]["host"] = "C (sgm_tiles_submit / sgm_tiles_finish, include/sgm_tiles.h)"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_tiles_in_process_c(args, ranks, WORKLOADS, golden_digests):
    """`ranks` pipelines of the C host as THREADS of this process on ONE GPU, connected by the library's local transport (device
    copies ordered by events stand in for the xGMI transfers).  Not a multi-GPU measurement: the schedule's own cost."""
    import threading
    import torch
    import soc_project_stereo_matching_amd as S
    from . import tiles
    from .tiling import tile_rows

    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    lead, B = args.tile_lead, max(1, args.batch or 1)
    digests = golden_digests(args.workload)
    get, seeds_of = _inputs(S, torch, w, h, d, seed, digests, B)
    keep_last, ring_frames = 2 * ranks, 2
    group = tiles.LocalGroup(ranks, 0)
    snaps, errors, times, slots_of = {}, [], [0.0, 0.0], [0] * ranks
    bar = threading.Barrier(ranks)

    def rank_main(r):
        try:
            tr = group.transport(r)
            pipe, slots_of[r] = _c_pipeline(S, tiles, 0, r, ranks, w, h, opt, B, lead, args.in_flight, tr)
            ring = torch.zeros((ring_frames, B, h, w), dtype=torch.float32, device="cuda:0")
            pipe.result_ring(ring.data_ptr(), ring_frames)
            _c_run(torch, pipe, get, max(args.warmup, 1))
            torch.cuda.synchronize()
            bar.wait()
            if r == 0:
                times[0] = time.perf_counter()
            _c_run(torch, pipe, get, args.steps)
            bar.wait()
            if r == 0:
                times[1] = time.perf_counter()
            snaps.update({f: m.clone() for f, m in _c_snaps(ring, ring_frames, r, ranks, args.steps, keep_last).items()})
            pipe.close()
            tr.close()
        except Exception as exc:                                      # noqa: BLE001
            errors.append((r, repr(exc)))
            bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    group.close()
    if errors:
        raise RuntimeError(f"in-process tile ranks failed: {errors}")
    elapsed = times[1] - times[0]
    n_ok, n_bad, n_unpinned = _verify(snaps, seeds_of, digests, B)
    line = _line(args, B, w, h, d, ranks, tile_rows, slots_of[0], elapsed, n_ok, n_bad, n_unpinned, keep_last,
                 f"{ranks} row tiles per frame driven by {ranks} threads of one process on ONE GPU (sgm_tiles_local transport: device "
                 "copies stand in for xGMI): the schedule's own cost, not a multi-GPU measurement", 1)
    line["config"]["mode"] = f"tiles, {ranks} in-process ranks on one GPU"
    line["config"]["host"] = "C (sgm_tiles_submit / sgm_tiles_finish, include/sgm_tiles.h)"
    print(json.dumps(line), flush=True)


def run_tile_rank_alone_c(args, spec, WORKLOADS):
    """`spec` = "r/N": rank r of an N-rank pipeline of the C host alone on this GPU, its exchanges moving nothing (tiles.NullTransport).
    A projection input (1 / max over the ranks = the rate N GPUs could reach with the exchanges fully hidden), NOT a multi-GPU
    measurement; there are no results to verify."""
    import torch
    import soc_project_stereo_matching_amd as S
    from . import tiles
    from .tiling import tile_rows
    r, n = (int(v) for v in spec.split("/"))
    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    lead, B = args.tile_lead, max(1, args.batch or 1)
    null = tiles.NullTransport()
    pipe, slots = _c_pipeline(S, tiles, 0, r, n, w, h, opt, B, lead, args.in_flight, null.struct)
    get, _ = _inputs(S, torch, w, h, d, seed, None, B)
    _c_run(torch, pipe, get, max(args.warmup, 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _c_run(torch, pipe, get, args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    pipe.close()
    print(json.dumps({"tile_rank_alone": spec, "host": "C", "lead": lead, "slots": slots, "workload": args.workload, "rows": tile_rows(h, n)[r],
                      "frames": args.steps * B, "frames_per_step": B, "ms_per_frame": round(el / (args.steps * B) * 1e3, 4),
                      "fps_if_this_rank_were_the_slowest": round(args.steps * B / el, 2),
                      "note": "one rank's share of an N-rank pipeline alone on one GPU, exchanges skipped: a projection input, not a result"}),
          flush=True)
