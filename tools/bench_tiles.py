#!/usr/bin/env python3
"""Row-tile mode over the ranks of one node: every frame is cut into WORLD_SIZE row tiles, one per GPU
(soc_project_stereo_matching_amd/tiling.py), hand-overs and the row gather over torch.distributed.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_tiles.py --frames 32 [--workload kitti_1242x375_d128_p8]

Backend nccl (= RCCL over xGMI) by default; SGM_BENCH_BACKEND=gloo stages the hand-overs through the host, which is
how the path is rehearsed on a box whose ranks share one GPU.  This is BASELINE.json configs[3] ("KITTI batch of 32
frames row-tiled across 8 GPUs"); bench.py measures the frame-sharded path, which needs no communication at all.
Rank 0 prints one JSON line; frame 0 is checked against the golden digest of the reference."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="kitti_1242x375_d128_p8", choices=sorted(WORKLOADS))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd.tiling import DeviceTileEngine, make_links, match_tiled, tile_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_rank = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("SGM_BENCH_BACKEND", "nccl")
    links = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        links = make_links(dist)

    w, h, d, seed, golden = WORKLOADS[args.workload]
    opt = S.default_option(d)
    eng = DeviceTileEngine(local_rank, w, h, opt, tile_rows(h, world)[rank])
    pairs = [S.synth_pair(w, h, d, seed + k) for k in range(2)]
    frames = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in pairs]
    torch.cuda.synchronize()

    def one(k):
        l, r = frames[k % len(frames)]
        return match_tiled(eng, rank, world, l, r, h, dist=dist if world > 1 else None, links=links)

    first = one(0)
    for k in range(1, args.warmup):
        one(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.frames):
        one(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        verified = None
        if golden:
            with open(os.path.join(ROOT, "tests", "golden", "cases.json")) as f:
                want = {c["name"]: c for c in json.load(f)["cases"]}[golden]["sha256"]["final"]
            verified = hashlib.sha256(first.cpu().numpy().tobytes()).hexdigest() == want
        print(json.dumps({"metric": "Mdisp/s", "value": round(w * h * d * 8 * args.frames / elapsed / 1e6, 1), "unit": "Mdisp/s",
                          "fps": round(args.frames / elapsed, 2), "ms_per_frame": round(elapsed / args.frames * 1e3, 4),
                          "n_gpus": world, "frames": args.frames, "scaling": "strong",
                          "config": {"workload": args.workload, "sharding": f"{world} row tiles per frame, boundary hand-over + row "
                                     f"all-gather over {backend}", "tile_rows": tile_rows(h, world)},
                          "verified_against_golden": verified}), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
