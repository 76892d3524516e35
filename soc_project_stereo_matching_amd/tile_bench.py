"""bench.py --mode tiles: every frame cut into WORLD_SIZE row tiles, one per GPU, frames in flight through the ranks.

The schedule is tiling.TilePipeline (boundary hand-overs + row gather as one grouped exchange per step; speckle + median
on the frame's owner rank).  A step is one frame; `value` = W*H*D*8*frames / elapsed (strong scaling: the same frames
whatever the number of GPUs).  Backend nccl (= RCCL over xGMI) moves device tensors; SGM_BENCH_BACKEND=gloo stages them
through the host, which is how the path is rehearsed on a box whose ranks share one GPU.

Verification: the last frames every rank owns are snapshotted on the device inside the timed region and hashed afterwards
against the digests the reference's own C produced for their seeds (tests/golden/bench_frames.json)."""
import hashlib
import json
import os
import time

import numpy as np

PATHS = 8
HBM_PEAK_GBS = 8000.0


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
    slots = max(world + 2, args.in_flight or 0)
    eng = DeviceSlotEngine(local_rank, w, h, opt, tile_rows(h, world)[rank], slots, host_staged=(world > 1 and backend != "nccl"))
    pipe = TilePipeline(eng, rank, world, h, dist=dist if world > 1 else None)

    digests = golden_digests(args.workload)
    n_pairs = max(1, min(4, len(digests))) if digests else 2
    pairs = [S.synth_pair(w, h, d, seed + k) for k in range(n_pairs)]
    frames = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in pairs]
    torch.cuda.synchronize()
    get = lambda f: frames[f % n_pairs]                              # noqa: E731

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

    n_ok = n_bad = n_unpinned = 0
    for f, t in sorted(snaps.items()):
        sd = seed + f % n_pairs
        if sd not in digests:
            n_unpinned += 1
        elif hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() == digests[sd]:
            n_ok += 1
        else:
            n_bad += 1
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
        cells = w * h * d
        dp = -(-d // 16) * 16
        value = cells * PATHS * args.steps / elapsed / 1e6
        moved = w * h * dp * 16                                      # planes written once + read once
        line = {
            "metric": "Mdisp/s (W*H*D*paths per second), fps beside it",
            "value": round(value, 1), "unit": "Mdisp/s", "fps": round(args.steps / elapsed, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/u16 integer min-plus (f32 sub-pixel tail)", "data": "synthetic",
            "config": {"workload": args.workload, "mode": "tiles", "width": w, "height": h, "disparity_range": d, "paths": PATHS,
                       "stages": "census+cost+aggregate8+wta+lrcheck+speckle+median", "frames": args.steps,
                       "tile_rows": tile_rows(h, world), "slots_per_rank": slots,
                       "sharding": f"{world} row tiles per frame; per step one grouped exchange per rank over {backend}: boundary path "
                                   "costs to both neighbours + finished rows to the frame's owner; speckle+median on the owner"},
            "roofline": {"bound": "hbm", "kernel": "whole frame (all kernels, all ranks)", "achieved": round(moved * args.steps / elapsed / 1e9, 1),
                         "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(moved * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world), 4),
                         "traffic": None, "algorithmic_bytes": "W*H*Dp*16 per frame (eight u8 planes written once and read once)"},
            "frames_verified": n_ok, "frames_mismatched": n_bad, "frames_without_reference_digest": n_unpinned,
            "verified_against_golden": (n_bad == 0 and n_ok > 0 and n_unpinned == 0),
            "verification": f"sha256 of the last {keep_last} frames of the timed region (snapshotted on their owner ranks) vs the "
                            "reference's own C, tests/golden/bench_frames.json",
            "cpu_baseline": None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
