#!/usr/bin/env python3
"""bench.py -- throughput of the SGM hot path on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over ONE BATCH of --batch frames: sgm_reset + sgm_match_device (the
device-pointer forms of the reference's SGM_Reset + SGM_Match; the Reset is part of every pass, SURVEY.md
Q14) on frames that are already resident in HBM when the timed region starts.  Every kernel of the
pipeline covers all frames of the batch in one launch (MI355X runs only ~4 kernels concurrently, so
frames are batched per launch rather than spread over many streams).  All stages run (census, cost,
8-path aggregation, WTA, LR check, speckle removal, median = the options of the reference's main.c).
Frames are independent units, so with N GPUs every rank processes its own K batches (weak scaling,
no data-path collective); `value` is the whole-job aggregate.

Headline workload = BASELINE.json configs[1]: KITTI 1242x375, D=128, 8 paths.
Metric: Mdisp/s = W*H*D*paths*frames / t / 1e6  (BASELINE.json "metric").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, D, seed, golden case whose digest pins frame 0)
    "kitti_1242x375_d128_p8": (1242, 375, 128, 0x5EED0002, "c2_kitti_1242x375_d128"),
    "cone_450x375_d64_p8": (450, 375, 64, 0x5EED0001, "c1_synth_450x375_d64"),
    "middlebury_2880x1988_d256_p8": (2880, 1988, 256, 0x5EED0003, None),
    "drivingstereo_1762x800_d192_p8": (1762, 800, 192, 0x5EED0005, None),
}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
PATHS = 8


def cpu_baseline(w, h, d, seed, budget_s=25.0):
    """The reference's C (oracle/_ref, built from /root/reference in the build container) timed on one
    host core on a bounded sample of the same workload; falls back to our C port of it."""
    from oracle.pyoracle import Oracle, Reference, default_option
    import soc_project_stereo_matching_amd as S
    opt = default_option(d)
    ref = Reference.for_shape(w, h, d)
    frames = [S.synth_pair(w, h, d, seed + k) for k in range(2)]
    if ref is not None:
        kind, run = "reference", (lambda l, r: ref.api_match(l, r, opt, reset=True))
    else:
        orc = Oracle()
        kind = "port"

        def run(l, r):
            assert orc.reset(w, h, opt)
            return orc.match(l, r)
    times = []
    t_start = time.perf_counter()
    k = 0
    while True:
        l, r = frames[k % 2]
        t0 = time.perf_counter()
        out = run(l, r)
        times.append(time.perf_counter() - t0)
        assert out is not None
        k += 1
        if k >= 5 or (time.perf_counter() - t_start) + times[-1] > budget_s:
            break
    t = float(np.median(times))
    return {"value": round(w * h * d * PATHS / t / 1e6, 2), "unit": "Mdisp/s", "cores": 1, "kind": kind,
            "sample": f"{k} frame(s) of {w}x{h} D={d} 8 paths, SGM_Reset+SGM_Match each, median; "
                      f"{t:.2f} s/frame = {1.0 / t:.3f} fps on 1 of {os.cpu_count()} host cores",
            "fps": round(1.0 / t, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8, help="frames per step (one launch per stage covers them all)")
    ap.add_argument("--workload", default="kitti_1242x375_d128_p8", choices=sorted(WORKLOADS))
    ap.add_argument("--in-flight", type=int, default=2, help="instances (HIP streams) a rank round-robins batches over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one process per GPU; if the launcher already narrowed the visible devices to one per rank, that one is device 0
    local_rank = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(local_rank)
    # nccl (= RCCL) on a multi-GPU node; SGM_BENCH_BACKEND=gloo rehearses the same multi-process path on a box whose
    # ranks share one GPU (RCCL refuses two ranks on one device)
    backend = os.environ.get("SGM_BENCH_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    w, h, d, seed, golden = WORKLOADS[args.workload]
    opt = S.default_option(d)
    n_inst = max(1, args.in_flight)
    B = max(1, args.batch)
    insts = [S.SGMInstance(local_rank, batch=B) for _ in range(n_inst)]
    for i in insts:
        if not i.reset(w, h, opt):
            raise SystemExit("sgm_reset failed")
        i.enable_timing(True)

    # synthetic batches, resident in HBM before the timed region (2 distinct batches of B distinct pairs per rank)
    n_frames = 2
    frames = []
    for k in range(n_frames):
        pairs = [S.synth_pair(w, h, d, seed + k * B + j + 4096 * rank) for j in range(B)]
        frames.append((torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(),
                       torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()))
    outs = [torch.empty((B, h, w), dtype=torch.float32, device="cuda") for _ in range(n_inst)]
    torch.cuda.synchronize()

    def step(k):
        i = insts[k % n_inst]
        l, r = frames[k % n_frames]
        if not i.reset(w, h, opt):                       # SGM_Reset: part of every frame (Q14)
            raise RuntimeError("sgm_reset failed")
        if not i.match_device(l.data_ptr(), r.data_ptr(), outs[k % n_inst].data_ptr()):
            raise RuntimeError("sgm_match_device failed")

    def barrier():
        if world > 1:
            dist.barrier()

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    for i in insts:
        i.enable_timing(True)                            # new statistics window: only the timed region is averaged
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    for i in insts:
        i.synchronize()                                  # collects the HIP-event stage times of each instance's last frame

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device time, mean over EVERY launch of the timed region (HIP events on the instance's own stream,
    # one event set per step; at most 64 steps per instance are kept)
    stage_sum, stage_min, launches = {}, {}, 0
    for i in insts:
        mean, mn, cnt = i.mean_timing()
        launches += cnt
        for name in mean:
            stage_sum[name] = stage_sum.get(name, 0.0) + mean[name] * cnt
            stage_min[name] = min(stage_min.get(name, 1e30), mn[name])
    stage_ms = {k: v / launches for k, v in stage_sum.items()} if launches else {}

    # single-frame latency: one batch-1 instance, nothing else in flight, after the timed region
    solo = S.SGMInstance(local_rank)
    lat = []
    for it in range(12):
        if it == 8:
            solo.enable_timing(True)                 # the last 4 frames give the stage breakdown (events cost ~5 us each)
        solo.reset(w, h, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        solo.match_device(frames[0][0].data_ptr(), frames[0][1].data_ptr(), outs[0].data_ptr())
        solo.synchronize()
        if 2 <= it < 8:
            lat.append(time.perf_counter() - t1)     # latency without the timing events
    solo_ms = solo.mean_timing()[0]
    first_frame = outs[0][0].cpu().numpy()           # frame 0 of batch 0 = the golden case's frame
    solo.close()

    verified = None
    if golden is not None and rank == 0:
        import hashlib
        with open(os.path.join(ROOT, "tests", "golden", "cases.json")) as f:
            want = {c["name"]: c for c in json.load(f)["cases"]}[golden]["sha256"]["final"]
        verified = hashlib.sha256(first_frame.tobytes()).hexdigest() == want

    if rank == 0:
        total_frames = args.steps * B * world
        cells = w * h * d
        value = cells * PATHS * total_frames / elapsed / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        # dominant kernel = the one-launch 8-direction aggregation: 5 algorithmic bytes per path
        # evaluation (read C 1 B + read-modify-write S 2+2 B) x W*H*D*8 per launch
        agg_ms = stage_ms.get("aggregate")
        agg_bytes = cells * 5 * PATHS * B               # one launch covers the B frames of a step
        roofline = None
        if agg_ms:
            achieved = agg_bytes / (agg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "sgm_aggregate_k", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": None, "avg_launch_ms": round(agg_ms, 4),
                        "min_launch_ms": round(stage_min["aggregate"], 4), "launches_timed": launches,
                        "algorithmic_bytes_per_launch": agg_bytes,
                        "note": "priced against the reference dataflow's 5 B per path evaluation (SURVEY.md 8d); the fused "
                                "kernel moves ~4.5x fewer bytes (traffic) and is VALU-issue bound, so frac can exceed 1"}
            # HBM bytes of this kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, measured at
            # one frame per launch, see profiles/hbm_traffic.json), scaled to the B frames of a launch
            prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(prof):
                with open(prof) as f:
                    per_frame = json.load(f).get(args.workload, {}).get("sgm_aggregate_k")
                if per_frame:
                    roofline["traffic"] = int(per_frame) * B
                    # the same launch priced with the bytes it really moves (what an HBM-bound kernel would be judged by)
                    roofline["traffic_GBps"] = round(roofline["traffic"] / (agg_ms * 1e-3) / 1e9, 1)
                    roofline["traffic_frac"] = round(roofline["traffic_GBps"] / HBM_PEAK_GBS, 4)
        # the other heavy kernel: cost sum + both WTAs, a pure HBM stream of the 8 planes (measured bytes, PMC)
        sum_roofline = None
        sum_ms = stage_ms.get("sum")
        if sum_ms and os.path.exists(os.path.join(ROOT, "profiles", "hbm_traffic.json")):
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                wl = json.load(f).get(args.workload, {})
            fused = wl.get("per_frame_fused_sum_wta", {}).get("sgm_sum_wta_lr_k")
            if fused and h * B >= 1024 and d <= 128:   # what the library's choice of the fused kernel needs
                nbytes = int(fused["hbm_bytes"]) * B
                sum_roofline = {"bound": "hbm", "kernel": "sgm_sum_wta_lr_k", "achieved": round(nbytes / (sum_ms * 1e-3) / 1e9, 1),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / (sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "traffic": nbytes, "avg_launch_ms": round(sum_ms, 4), "min_launch_ms": round(stage_min["sum"], 4),
                                "note": "bytes = measured HBM traffic (8 planes read once); with a second batch in flight the "
                                        "launch shares HBM with the other batch's aggregation"}
        frame_bytes = cells * (5 * PATHS + 3)
        steps_frames = args.steps * B
        line = {
            "metric": "Mdisp/s (W*H*D*paths per second), fps beside it",
            "value": round(value, 1), "unit": "Mdisp/s",
            "fps": round(total_frames / elapsed, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u16 integer min-plus (f32 sub-pixel tail)", "data": "synthetic",
            "config": {"workload": args.workload, "width": w, "height": h, "disparity_range": d, "paths": PATHS,
                       "stages": "census+cost+aggregate8+wta+lrcheck+speckle+median", "frames_per_step": B,
                       "frames_per_gpu": args.steps * B, "batches_in_flight_per_gpu": n_inst,
                       "sharding": "independent frames per rank, no collective"},
            "roofline": roofline,
            "roofline_sum_wta": sum_roofline,
            "frame_roofline": {"algorithmic_bytes_per_frame": frame_bytes,
                               "achieved_GBps_per_gpu": round(frame_bytes * steps_frames / elapsed / 1e9, 1),
                               "frac_per_gpu": round(frame_bytes * steps_frames / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
            "ms_per_frame": round(elapsed / steps_frames * 1e3, 4),
            "stage_ms_per_batch_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_single_frame": {k: round(v, 4) for k, v in solo_ms.items()},
            "single_frame_latency_ms": round(float(np.median(lat)) * 1e3, 4),
            "verified_against_golden": verified,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(w, h, d, seed)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    for i in insts:
        i.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
