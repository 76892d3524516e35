#!/usr/bin/env python3
"""bench.py -- throughput of the SGM hot path on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W [--mode tiles]

--mode frames (default).  A "step" is one pass of the hot path over ONE BATCH of --batch frames: sgm_reset +
sgm_match_device (the device-pointer forms of the reference's SGM_Reset + SGM_Match; the Reset is part of every pass,
SURVEY.md Q14) on frames that are already resident in HBM when the timed region starts.  Every kernel of the pipeline
covers all frames of the batch in one launch.  All stages run (census, cost, 8-path aggregation, WTA, LR check,
speckle removal, median = the options of the reference's main.c).  Frames are independent units, so with N GPUs every
rank processes its own K batches (weak scaling, no data-path collective); `value` is the whole-job aggregate.

--mode tiles.  Every frame is cut into N row tiles, one per GPU (soc_project_stereo_matching_amd/tiling.py): boundary
path costs are handed from rank to rank, several frames are in flight so that the ranks work as a pipeline, the rows are
gathered on the frame's owner rank, which runs speckle removal + median.  A step is one frame (--batch B: the same tile of B
frames per launch and per hand-over); strong scaling.  On ONE GPU: --tile-ranks-in-process N runs N ranks as threads (device copies
for the hand-overs), --tile-rank-alone r/N one rank's share with the exchanges skipped (tools/tiles_schedule_cost.py).

Every frame of the LAST timed batch of every in-flight instance is hashed against the digest the reference's own C
produced for that seed (tests/golden/bench_frames.json): `frames_verified`.

Headline workload = BASELINE.json configs[1]: KITTI 1242x375, D=128, 8 paths.
Metric: Mdisp/s = W*H*D*paths*frames / t / 1e6  (BASELINE.json "metric").
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (W, H, D, first seed); frame f of a rank's stream has seed first + f (digests: tests/golden/bench_frames.json)
    "kitti_1242x375_d128_p8": (1242, 375, 128, 0x5EED0002),
    "cone_450x375_d64_p8": (450, 375, 64, 0x5EED0001),
    "middlebury_2880x1988_d256_p8": (2880, 1988, 256, 0x5EED0003),
    "drivingstereo_1762x800_d192_p8": (1762, 800, 192, 0x5EED0005),
    "uhd_3840x2160_d128_p8": (3840, 2160, 128, 0x5EED0006),
    "uhd_3840x2160_d256_p8": (3840, 2160, 256, 0x5EED0007),
}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9               # max shader clock
VALU_ISSUE_CYCLES = 4          # cycles one wave-level VALU instruction occupies its SIMD's issue (profiles/*valu_rate*.txt)
PATHS = 8


def source_id():
    """Identifies the kernels a counter profile belongs to: sha256 over the HIP sources (first 16 hex digits)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def golden_digests(workload):
    path = os.path.join(ROOT, "tests", "golden", "bench_frames.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        wl = json.load(f)["workloads"].get(workload, {})
    return {int(k): v["sha256"]["final"] for k, v in wl.get("frames", {}).items()}


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


def cpu_baseline_all_cores(w, h, d, seed, max_procs=16, timeout_s=90.0):
    """Whole-host figure beside the one-core baseline: one independent process per available host core (at most 16 = a
    one-GPU box's CPU share), each running tools/cpu_frame.py (one warm-up frame, one timed frame through the reference's C;
    separate processes because the reference keeps its state in globals), all at once; throughput = processes / slowest
    frame time.  Plain child processes with a wall-clock limit: this leg can fail, it cannot hang the bench."""
    import subprocess
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, max_procs))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_frame.py"), str(w), str(h), str(d)]
    procs = [subprocess.Popen(cmd + [str(seed + k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(n)]
    deadline = time.perf_counter() + timeout_s
    times = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.perf_counter()))
            times.append(float(out.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    t = max(times)
    return {"value": round(w * h * d * PATHS * n / t / 1e6, 2), "unit": "Mdisp/s", "cores": n, "fps": round(n / t, 3),
            "sample": f"{n} processes x 1 frame of {w}x{h} D={d} at the same time (after a warm-up frame each); slowest {t:.2f} s"}


def host_boundary(S, device, w, h, d, opt, pairs, B, digests, seeds, budget_s=6.0):
    """What a caller of the HOST-pointer boundary gets (PCIe inclusive; never `value`).
    blocking  = the reference contract: SGM_Reset + SGM_Match per frame on pageable arrays, one frame at a time.
    pipelined = sgm_reset + sgm_match_async on batches of B frames round-robined over 4 instances, each driven by
                its own host thread (the staging copies of one batch overlap the kernels of the others), once with
                pageable caller buffers (staged through pinned memory) and once with sgm_host_alloc'ed buffers."""
    res = {}
    # ---- blocking, one frame per call, global reference entry points
    g = S.SGM()
    l0, r0 = pairs[0]
    n, t_sum, ok = 0, 0.0, True
    t_begin = time.perf_counter()
    while n < 200 and time.perf_counter() - t_begin < budget_s / 3:
        l, r = pairs[n % len(pairs)]
        t0 = time.perf_counter()
        out = g.compute(l, r, opt)                               # sgm_compute = SGM_Reset + SGM_Match
        dt = time.perf_counter() - t0
        if n >= 2:
            t_sum += dt
        if out is None:
            ok = False
            break
        n += 1
    if ok and n > 2:
        ms = t_sum / (n - 2) * 1e3
        sd = seeds[(n - 1) % len(pairs)]
        res["blocking_single_frame"] = {"ms_per_frame": round(ms, 4), "fps": round(1e3 / ms, 1), "frames": n - 2,
                                        "entry": "sgm_compute (SGM_Reset + SGM_Match), pageable numpy arrays",
                                        "verified": (hashlib.sha256(out.tobytes()).hexdigest() == digests[sd]) if sd in digests else None}
    g.shutdown()

    # ---- pipelined: 4 instances x batch B, one host thread each (3: 0.81 / 0.95 of the device-resident rate, 4: 0.95 / 0.97)
    for kind in ("pageable", "pinned"):
        n_inst = int(os.environ.get("SGM_BENCH_HOST_INSTANCES", "4"))
        insts = [S.SGMInstance(device, batch=B) for _ in range(n_inst)]
        bufs = []
        for i in insts:
            if int(os.environ.get("SGM_BENCH_HOST_OVERLAP_POST", "1")):
                i.set_overlap_post(True)             # a result is handed over by sgm_match_wait anyway: the second stream is free here
            assert i.reset(w, h, opt)
            if kind == "pinned":
                L, R, O = i.host_array((B, h, w), np.uint8), i.host_array((B, h, w), np.uint8), i.host_array((B, h, w), np.float32)
            else:
                L, R, O = np.empty((B, h, w), np.uint8), np.empty((B, h, w), np.uint8), np.empty((B, h, w), np.float32)
            for j in range(B):
                L[j], R[j] = pairs[j % len(pairs)]
            bufs.append((L, R, O))
        rounds = [0] * n_inst
        stop_at = [0.0]
        fail = []

        def worker(k):
            i, (L, R, O) = insts[k], bufs[k]
            while time.perf_counter() < stop_at[0]:
                if not (i.reset(w, h, opt) and i.match_async(L, R, O) and i.match_wait()):
                    fail.append(k)
                    return
                rounds[k] += 1

        # warm-up round, then the timed window
        for k in range(n_inst):
            i, (L, R, O) = insts[k], bufs[k]
            assert i.reset(w, h, opt) and i.match_async(L, R, O) and i.match_wait()
        t0 = time.perf_counter()
        stop_at[0] = t0 + budget_s / 3
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_inst)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        frames = sum(rounds) * B
        ver = None
        if not fail and all(seeds[j % len(pairs)] in digests for j in range(B)):
            ver = all(hashlib.sha256(bufs[k][2][j].tobytes()).hexdigest() == digests[seeds[j % len(pairs)]]
                      for k in range(n_inst) for j in range(B))
        res[f"pipelined_{kind}"] = {"fps": round(frames / el, 1), "ms_per_frame": round(el / max(frames, 1) * 1e3, 4), "frames": frames,
                                    "instances": n_inst, "frames_per_call": B, "host_threads": n_inst,
                                    "entry": "sgm_reset + sgm_match_async + sgm_match_wait", "verified": ver, "failed": bool(fail)}
        for i in insts:
            i.close()
    return res


def load_counters(workload):
    """Per-kernel hardware counters collected by tools/profile_counters.py (rocprofv3 --pmc passes), stamped with
    the source id of the kernels they were measured on."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        doc = json.load(f)
    wl = doc.get("workloads", {}).get(workload)
    if wl is None:
        return None
    return {"source_id": doc.get("source_id"), "file": "profiles/counters.json", **wl}


def kernel_roofline(kernel, ms, launches, min_ms, B, counters, alg_bytes_per_frame, alg_note, ref_equiv_bytes_per_frame=None):
    """roofline object of one kernel.  `achieved` / `frac` = HBM bytes the launch really moves (`traffic`, from the PMC passes
    of profiles/counters.json) / mean launch time (HIP events over the timed region) against the 8 TB/s peak; where no fresh
    counters exist they fall back to the bytes this dataflow has to move (`algorithmic_*`, always reported beside them).
    `valu` = share of the chip's VALU issue slots the launch's wave-level VALU instructions take.  `bound` names the larger
    of the two fractions."""
    t = ms * 1e-3
    alg = alg_bytes_per_frame * B
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(alg / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4), "bytes_used": "algorithmic", "traffic": None, "avg_launch_ms": round(ms, 4),
         "min_launch_ms": round(min_ms, 4), "launches_timed": launches, "frames_per_launch": B,
         "algorithmic_bytes_per_launch": int(alg), "algorithmic_bytes": alg_note,
         "algorithmic_GBps": round(alg / t / 1e9, 1), "algorithmic_frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4)}
    if counters:
        k = counters.get("kernels", {}).get(kernel)
        stale = counters.get("source_id") != source_id()
        r["counters_file"] = counters.get("file")
        r["traffic_stale"] = stale
        if k:
            if k.get("hbm_bytes_per_frame") is not None:
                r["traffic"] = int(k["hbm_bytes_per_frame"] * B)
                if not stale:
                    r["achieved"] = round(r["traffic"] / t / 1e9, 1)
                    r["frac"] = round(r["traffic"] / t / 1e9 / HBM_PEAK_GBS, 4)
                    r["bytes_used"] = "measured traffic (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/counters.json)"
            if k.get("valu_insts_per_frame") is not None:
                insts = k["valu_insts_per_frame"] * B
                vfrac = insts * VALU_ISSUE_CYCLES / (N_SIMD * CLOCK_HZ * t)
                r["valu"] = {"wave_insts_per_launch": int(insts), "issue_cycles_per_inst": VALU_ISSUE_CYCLES,
                             "frac": round(vfrac, 4), "of": f"{N_SIMD} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz"}
                if vfrac > r["frac"]:
                    r["bound"] = "valu"
    if ref_equiv_bytes_per_frame:
        eq = ref_equiv_bytes_per_frame * B / t / 1e9
        r["reference_dataflow_equiv"] = {"bytes_per_launch": int(ref_equiv_bytes_per_frame * B), "GBps": round(eq, 1),
                                         "x_peak": round(eq / HBM_PEAK_GBS, 3),
                                         "note": "the reference dataflow's 5 B per path evaluation (SURVEY.md 8d) that this launch replaces; "
                                                 "not a fraction of anything physical (this kernel moves ~4.5x fewer bytes)"}
    return r


def init_dist(args):
    import torch
    import torch.distributed as dist
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
    return world, rank, local_rank, backend


def run_frames(args):
    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S

    world, rank, local_rank, backend = init_dist(args)
    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    n_inst = max(1, args.in_flight)
    B = max(1, args.batch)
    insts = [S.SGMInstance(local_rank, batch=B) for _ in range(n_inst)]
    overlap_post = args.overlap_post if args.overlap_post is not None else int(os.environ.get("SGM_BENCH_OVERLAP_POST", "0"))
    cu_split = args.cu_split if args.cu_split is not None else os.environ.get("SGM_BENCH_CU_SPLIT", "")
    for i in insts:
        if overlap_post and not i.set_overlap_post(True):
            raise SystemExit("sgm_set_overlap_post failed")
        if cu_split and not i.set_cu_split(cu_split):
            raise SystemExit(f"sgm_set_stage_cus failed for {cu_split!r}")
        if not i.reset(w, h, opt):
            raise SystemExit("sgm_reset failed")
        i.enable_timing(True)

    # synthetic batches, resident in HBM before the timed region: 2 distinct batches of B distinct pairs, the same on
    # every rank (every rank can then check its results against the reference's digests)
    n_frames = 2
    frames, pairs, seeds = [], [], []
    for k in range(n_frames):
        ps = [S.synth_pair(w, h, d, seed + k * B + j) for j in range(B)]
        pairs += ps
        seeds += [seed + k * B + j for j in range(B)]
        frames.append((torch.from_numpy(np.stack([p[0] for p in ps])).cuda(),
                       torch.from_numpy(np.stack([p[1] for p in ps])).cuda()))
    outs = [torch.empty((B, h, w), dtype=torch.float32, device="cuda") for _ in range(n_inst)]
    last_batch = [None] * n_inst                         # which batch an instance's output buffer holds
    torch.cuda.synchronize()

    def step(k):
        i = insts[k % n_inst]
        l, r = frames[k % n_frames]
        if not i.reset(w, h, opt):                       # SGM_Reset: part of every frame (Q14)
            raise RuntimeError("sgm_reset failed")
        if not i.match_device(l.data_ptr(), r.data_ptr(), outs[k % n_inst].data_ptr()):
            raise RuntimeError("sgm_match_device failed")
        last_batch[k % n_inst] = k % n_frames

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
        step(args.warmup + k)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    for i in insts:
        i.synchronize()                                  # collects the HIP-event stage times of each instance's last frame

    # ---- verify what was timed: every frame of the last batch each in-flight instance produced, against the digests
    #      of the reference's own C for those seeds (tests/golden/bench_frames.json)
    digests = golden_digests(args.workload)
    n_ok = n_bad = n_unpinned = 0
    for k in range(n_inst):
        if last_batch[k] is None:
            continue
        got = outs[k].cpu().numpy()
        for j in range(B):
            sd = seed + last_batch[k] * B + j
            if sd not in digests:
                n_unpinned += 1
            elif hashlib.sha256(got[j].tobytes()).hexdigest() == digests[sd]:
                n_ok += 1
            else:
                n_bad += 1
    if world > 1:
        t = torch.tensor([elapsed, float(n_ok), float(n_bad), float(n_unpinned)], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        n_ok, n_bad, n_unpinned = int(t[1].item()), int(t[2].item()), int(t[3].item())

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

    # the same batches with ONE instance and nothing else on the GPU (after the timed region): the kernels' own launch times,
    # without a second batch competing for VALU issue and HBM
    alone_ms = {}
    if n_inst > 1 and args.alone:
        for rep in range(2):                             # a first pass to settle, the second one is read
            insts[0].enable_timing(True)
            for k in range(10):
                l, r = frames[k % n_frames]
                if not (insts[0].reset(w, h, opt) and insts[0].match_device(l.data_ptr(), r.data_ptr(), outs[0].data_ptr())):
                    raise RuntimeError("sgm_match_device failed")
            insts[0].synchronize()
        alone_ms = insts[0].mean_timing()[0]

    # single-frame latency: one batch-1 instance, nothing else in flight, after the timed region (its own output buffer)
    solo = S.SGMInstance(local_rank)
    solo_out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    lat = []
    for it in range(12):
        if it == 8:
            solo.enable_timing(True)                 # the last 4 frames give the stage breakdown (events cost ~5 us each)
        solo.reset(w, h, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        solo.match_device(frames[0][0].data_ptr(), frames[0][1].data_ptr(), solo_out.data_ptr())
        solo.synchronize()
        if 2 <= it < 8:
            lat.append(time.perf_counter() - t1)     # latency without the timing events
    solo_ms = solo.mean_timing()[0]
    solo_ok = None
    if seed in digests:
        solo_ok = hashlib.sha256(solo_out.cpu().numpy().tobytes()).hexdigest() == digests[seed]
    solo.close()
    for i in insts:
        i.close()

    if rank == 0:
        total_frames = args.steps * B * world
        cells = w * h * d
        value = cells * PATHS * total_frames / elapsed / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        counters = load_counters(args.workload)
        dp = -(-d // 16) * 16
        # dominant kernel = the one-launch 8-direction aggregation.  Bytes this dataflow has to move per frame: the 8
        # per-direction L_r planes written once (8 B per cell of the padded volume) + both census images and the left
        # image read once (9 B per pixel).  (Reference dataflow: 5 B per path evaluation = 40 B per cell)
        roofline = None
        if stage_ms.get("aggregate"):
            roofline = kernel_roofline("sgm_aggregate_k", stage_ms["aggregate"], launches, stage_min["aggregate"], B, counters,
                                       w * h * dp * 8 + w * h * 9,
                                       "W*H*Dp*8 (eight u8 L_r planes written once) + W*H*9 (census L/R + left image read once)",
                                       ref_equiv_bytes_per_frame=cells * 5 * PATHS)
        sum_roofline = None
        if stage_ms.get("sum") and dp <= 256:
            sum_roofline = kernel_roofline("sgm_sum_wta_lr_k", stage_ms["sum"], launches, stage_min["sum"], B, counters,
                                           w * h * dp * 8 + w * h * 8,
                                           "W*H*Dp*8 (eight planes read once) + W*H*8 (two disparity maps written)")
        # ... and the same two objects for the launch times with the GPU to itself
        for rf, key in ((roofline, "aggregate"), (sum_roofline, "sum")):
            if rf and alone_ms.get(key):
                ta = alone_ms[key] * 1e-3
                alone = {"avg_launch_ms": round(alone_ms[key], 4), "batches_in_flight": 1}
                if rf.get("traffic"):
                    alone["frac"] = round(rf["traffic"] / ta / 1e9 / HBM_PEAK_GBS, 4)
                alone["algorithmic_frac"] = round(rf["algorithmic_bytes_per_launch"] / ta / 1e9 / HBM_PEAK_GBS, 4)
                if rf.get("valu"):
                    alone["valu_frac"] = round(rf["valu"]["wave_insts_per_launch"] * VALU_ISSUE_CYCLES / (N_SIMD * CLOCK_HZ * ta), 4)
                rf["alone"] = alone
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
            "config": {"workload": args.workload, "mode": "frames", "width": w, "height": h, "disparity_range": d, "paths": PATHS,
                       "stages": "census+cost+aggregate8+wta+lrcheck+speckle+median", "frames_per_step": B,
                       "frames_per_gpu": args.steps * B, "batches_in_flight_per_gpu": n_inst,
                       "post_pass_on_second_stream": bool(overlap_post) or "post" in cu_split,
                       "stage_cus_per_xcd": cu_split or None,
                       "sharding": "independent frames per rank, no collective"},
            "roofline": roofline,
            "roofline_sum_wta": sum_roofline,
            "frame_reference_dataflow_equiv": {"bytes_per_frame": frame_bytes,
                                               "GBps_per_gpu": round(frame_bytes * steps_frames / elapsed / 1e9, 1),
                                               "note": "W*H*D*(5*8+3) B of the reference dataflow (SURVEY.md 8d) per frame time; not a roofline fraction"},
            "ms_per_frame": round(elapsed / steps_frames * 1e3, 4),
            "stage_ms_per_batch_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_single_frame": {k: round(v, 4) for k, v in solo_ms.items()},
            "single_frame_latency_ms": round(float(np.median(lat)) * 1e3, 4),
            "frames_verified": n_ok, "frames_mismatched": n_bad, "frames_without_reference_digest": n_unpinned,
            "verified_against_golden": (n_bad == 0 and n_ok > 0 and n_unpinned == 0),
            "verification": "sha256 of every frame of the last timed batch of each in-flight instance (all ranks) vs the "
                            "reference's own C for the same seeds, tests/golden/bench_frames.json",
            "single_frame_verified": solo_ok,
            "source_id": source_id(),
        }
        if counters and counters.get("kernels"):
            tot = sum(v.get("hbm_bytes_per_frame") or 0 for v in counters["kernels"].values())
            if tot:
                line["frame_traffic"] = {"hbm_bytes_per_frame": int(tot), "GBps_per_gpu": round(tot * steps_frames / elapsed / 1e9, 1),
                                         "frac_of_peak": round(tot * steps_frames / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                                         "stale": counters.get("source_id") != source_id()}
        if world == 1 and not args.no_host_boundary:
            hb = host_boundary(S, local_rank, w, h, d, opt, pairs, B, digests, seeds)
            for v in hb.values():
                v["vs_device_resident"] = round(v["fps"] / line["fps"], 3)
            line["host_boundary"] = hb
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(w, h, d, seed)
            try:
                cb["all_cores"] = cpu_baseline_all_cores(w, h, d, seed)
            except Exception as e:                                   # the one-core figure is the contract; this one is extra
                cb["all_cores"] = {"error": repr(e)}
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--mode", default="frames", choices=["frames", "tiles"])
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per step, one launch per stage covers them all (frames mode: default 8; tiles mode: default 1 = "
                         "one frame cut into the ranks' tiles per step, N = the same tile of N frames per launch)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--in-flight", type=int, default=None,
                    help="frames mode: instances (HIP streams) a rank round-robins batches over (default 2); "
                         "tiles mode: frames in flight through the rank pipeline (default 2 x ranks)")
    ap.add_argument("--tile-ranks-in-process", type=int, default=0,
                    help="tiles mode on ONE GPU: this many tile ranks as threads of one process (device copies stand in for xGMI): "
                         "what the pipeline schedule itself costs against --mode tiles with one rank; not a multi-GPU measurement")
    ap.add_argument("--tile-lead", type=int, default=int(os.environ.get("SGM_TILE_LEAD", "2")),
                    help="tiles mode: steps tile_begin of a frame is queued ahead of its first sweep (tiling.TilePipeline lead)")
    ap.add_argument("--tile-rank-alone", default=None, metavar="r/N",
                    help="tiles mode: rank r of an N-rank pipeline alone on this GPU with the exchanges skipped: its time per frame "
                         "(a projection input for N GPUs, results are not produced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true")
    ap.add_argument("--overlap-post", type=int, default=None, choices=[0, 1],
                    help="sgm_set_overlap_post on the bench's instances: LR check / speckle / median of a batch on a second stream "
                         "beside the next batch's aggregation (default: SGM_BENCH_OVERLAP_POST or 0: +1..3 %% fps with two batches in "
                         "flight, +7 %% with one, but the launches then overlap more and the per-launch roofline fractions read lower)")
    ap.add_argument("--cu-split", default=None, metavar="SPEC",
                    help="sgm_set_stage_cus on the bench's instances: stage groups on streams and compute units of their own, e.g. "
                         "'post=0:2,sum=2:8,main=10:22' = first:count CUs of every XCD (count 0: own stream on all CUs)")
    ap.add_argument("--alone", action="store_true",
                    help="after the timed region also time the batches with ONE instance and nothing else on the GPU -> roofline.alone "
                         "(off by default: a kernel trace of the default run then holds the timed configuration only, so rocprofv3's "
                         "per-kernel average and roofline.avg_launch_ms describe the same launches)")
    args = ap.parse_args()
    if args.mode == "tiles":
        # a rank of the tile pipeline keeps ~N + 3 HIP streams busy, some with millisecond-long serial kernels (tile_begin's
        # horizontal lines); on HIP's default of 4 hardware queues short kernels queue up behind those (measured, DESIGN.md
        # section 7: 1.9 -> 1.45 ms per frame for one rank of eight at 3840x2160).  Read by the runtime when it starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        from soc_project_stereo_matching_amd.tile_bench import run_tiles, run_tiles_in_process
        args.workload = args.workload or "uhd_3840x2160_d128_p8"
        if args.tile_rank_alone:
            from soc_project_stereo_matching_amd.tile_bench import run_tile_rank_alone
            return run_tile_rank_alone(args, args.tile_rank_alone, WORKLOADS)
        if args.tile_ranks_in_process > 1:
            return run_tiles_in_process(args, args.tile_ranks_in_process, WORKLOADS, golden_digests)
        return run_tiles(args, init_dist, WORKLOADS, golden_digests)
    if args.cu_split or os.environ.get("SGM_BENCH_CU_SPLIT"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # up to three streams per instance in flight: no two of them on one hardware queue
    args.workload = args.workload or "kitti_1242x375_d128_p8"
    args.batch = args.batch or 8
    args.in_flight = args.in_flight or 2
    run_frames(args)


if __name__ == "__main__":
    main()
