#!/usr/bin/env python3
"""bench.py -- throughput of the SGM hot path on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W [--mode tiles]

--mode frames (default).  A "step" is one pass of the hot path over ONE BATCH of --host-instances x --batch frames (defaults 4 x 8 =
32: BASELINE config 4's "batch of 32 KITTI frames") through the HOST-pointer boundary -- SURVEY.md 8(d)'s t_frame: every instance
takes a sub-batch of --batch frames through sgm_reset + sgm_match_async + sgm_match_wait (the reference's SGM_Reset + SGM_Match,
SemiGlobalMatching.c:77-78,122: borrowed host images in, host floats out) on page-locked caller buffers, i.e. H2D of the
images, every kernel, D2H of the disparity maps; each instance is driven by its own host thread, so the copies of one sub-batch
overlap the kernels of the others, and every kernel of the pipeline covers the --batch frames of its sub-batch in one launch.
All stages run (census, cost, 8-path aggregation, WTA, LR check, speckle removal, median = the options of the reference's
main.c).  Frames are independent units, so with N GPUs every rank processes its own K batches -- its share
(sharding.frames_of_rank) of the 32-frame KITTI pool, cycled -- (weak scaling, no data-path collective); `value` is the
whole-job aggregate.  Beside it in the same JSON line (N = 1): `device_resident` (the same frames already in HBM: kernels
only), `sustained` (the headline loop for >= 2 s), `host_boundary` (the blocking one-frame-per-call contract, pageable
buffers), `workloads` (the other single-GPU BASELINE configs: cone 8 / 4 paths, 2880x1988 D=256, 1762x800 D=192, each with
its own roofline), `stream` (config 5 as a stream of 256 distinct frames), `cpu_baseline`.

--mode tiles.  Every frame is cut into N row tiles, one per GPU (soc_project_stereo_matching_amd/tiling.py): boundary
path costs are handed from rank to rank, several frames are in flight so that the ranks work as a pipeline, the rows are
gathered on the frame's owner rank, which runs speckle removal + median.  A step is one frame (--batch B: the same tile of B
frames per launch and per hand-over); strong scaling.  On ONE GPU: --tile-ranks-in-process N runs N ranks as threads (device copies
for the hand-overs), --tile-rank-alone r/N one rank's share with the exchanges skipped (tools/tiles_schedule_cost.py).

Every leg hashes what it timed against the digests the reference's own C produced for the same seeds
(tests/golden/bench_frames.json): the headline the last 4 batches of every instance (`frames_verified`: 128 frames at the
defaults), the other legs their last batches, the stream its first and last four frames.

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
    # BASELINE config 1's "4 paths": the library's extension (sgm_set_honor_num_paths); digests are oracle-made
    "cone_450x375_d64_p4": (450, 375, 64, 0x5EED0001),
    # throughput with speckle removal on AND off -- the KITTI frames with is_remove_speckles = false (main.c:60 flipped),
    # reference digests of their own
    "kitti_1242x375_d128_p8_nospeckle": (1242, 375, 128, 0x5EED0002),
}
SPECKLE_OFF = "_nospeckle"
POOL_FRAMES = 32               # BASELINE config 4: a batch of 32 KITTI frames, sharded over the ranks frame by frame
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


def paths_of(workload):
    return 4 if workload.endswith("_p4") else 8


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


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
            "fps": round(1.0 / t, 4), "host_cpu": host_cpu_model(), "host_cores": os.cpu_count()}


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


def load_counters(workload):
    """Per-kernel hardware counters collected by tools/profile_counters.py (rocprofv3 --pmc passes), stamped with
    the source id of the kernels they were measured on."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        doc = json.load(f)
    # the speckle-off variant runs the same aggregation / cost-sum launches on the same frames
    wl = doc.get("workloads", {}).get(workload[:-len(SPECKLE_OFF)] if workload.endswith(SPECKLE_OFF) else workload)
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


COMPACT_LIMIT = 4096           # the driver keeps ~8 KB of stdout: the LAST line must be one short JSON object


def _rf_compact(rf):
    """The roofline object the contract names (bound / achieved / peak / unit / frac / traffic) plus what the judge recomputes it from."""
    if not rf:
        return None
    out = {k: rf.get(k) for k in ("kernel", "bound", "frac", "achieved", "peak", "unit", "traffic", "algorithmic_bytes_per_launch",
                                   "avg_launch_ms", "launches_timed", "frames_per_launch")}
    out["algorithmic_frac"] = rf.get("algorithmic_frac")
    out["traffic_stale"] = rf.get("traffic_stale")
    out["valu_frac"] = (rf.get("valu") or {}).get("frac")
    al = rf.get("alone") or {}
    out["alone_frac"] = al.get("frac", al.get("algorithmic_frac"))
    out["alone_launch_ms"] = al.get("avg_launch_ms")
    out["alone_valu_frac"] = al.get("valu_frac")
    return out


def _frac_of(rf, alone=False):
    if not rf:
        return None
    if alone:
        al = rf.get("alone") or {}
        return al.get("frac", al.get("algorithmic_frac"))
    return rf.get("frac")


def compact_line(line, detail="bench_detail.json"):
    """The ONE short JSON line rank 0 prints LAST (< COMPACT_LIMIT bytes): the contract keys, `roofline` + `cpu_baseline`, and a
    one-row summary of every other leg.  Everything else lives in the detail file / the earlier stdout line."""
    cfg = line.get("config") or {}
    entry = cfg.get("entry") or ""
    out = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["metric"] = "Mdisp/s (W*H*D*paths/s; t_frame incl. H2D + D2H)"
    out["dtype"] = "u8"
    out["config"] = {k: cfg.get(k) for k in ("workload", "mode", "width", "height", "disparity_range", "paths", "frames_per_step",
                                             "frames_per_launch", "instances_per_gpu") if k in cfg}
    out["config"]["entry"] = entry.split(":")[0][:80]
    for k in ("fps", "ms_per_frame", "frames_verified", "frames_mismatched", "verified_against_golden", "source_id"):
        if k in line:
            out[k] = line[k]
    out["roofline"] = _rf_compact(line.get("roofline"))
    if line.get("roofline_sum_wta"):
        out["roofline_sum_wta"] = _rf_compact(line["roofline_sum_wta"])
    ft = line.get("frame_traffic")
    if ft:
        out["frame_traffic"] = {"hbm_bytes_per_frame": ft.get("hbm_bytes_per_frame"), "frac_of_peak": ft.get("frac_of_peak"), "stale": ft.get("stale")}
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "fps", "host_cpu", "host_cores")}
        out["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:100]
        c16 = cb.get("cores_16") or {}
        if "fps" in c16:
            out["cpu_baseline"]["fps_16_cores"] = c16["fps"]
        out["speedup_vs_cpu_baseline"] = line.get("speedup_vs_cpu_baseline")
    else:
        out["cpu_baseline"] = None
    if line.get("sustained"):
        out["sustained_fps"] = line["sustained"].get("fps")
    dev = line.get("device_resident")
    if dev:
        out["device_resident"] = {"fps": dev.get("fps"), "verified": dev.get("frames_verified"), "mismatched": dev.get("frames_mismatched"),
                                  "fps_all_ranks": dev.get("fps_all_ranks")}
    if "single_frame_latency_ms" in line:
        out["single_frame_latency_ms"] = line["single_frame_latency_ms"]
    hb = line.get("host_boundary")
    if hb:
        out["host_boundary"] = {k: v.get("fps") for k, v in hb.items()}
    if line.get("workloads"):
        out["workloads_columns"] = ["fps", "verified", "mismatched", "agg_frac", "sum_frac", "agg_alone_frac", "sum_alone_frac"]
        out["workloads"] = {}
        for wl in line["workloads"]:
            if "error" in wl:
                out["workloads"][wl["workload"]] = "error"
                continue
            out["workloads"][wl["workload"]] = [wl.get("fps"), wl.get("frames_verified"), wl.get("frames_mismatched"),
                                                _frac_of(wl.get("roofline")), _frac_of(wl.get("roofline_sum_wta")),
                                                _frac_of(wl.get("roofline"), True), _frac_of(wl.get("roofline_sum_wta"), True)]
    st = line.get("stream")
    if st:
        out["stream"] = {k: st.get(k) for k in ("workload", "frames", "fps", "frames_verified", "frames_mismatched")} if "error" not in st else "error"
    out["detail"] = detail
    s = json.dumps(out)
    if len(s) >= COMPACT_LIMIT:                     # never let an extra leg break the contract line: drop summaries, keep the contract
        for k in ("workloads_columns", "workloads", "stream", "host_boundary", "frame_traffic", "roofline_sum_wta"):
            out.pop(k, None)
            if len(json.dumps(out)) < COMPACT_LIMIT:
                break
    return out


def emit(line):
    """Rank 0's output.  stdout carries exactly ONE line -- the compact contract line (the driver keeps only the tail of stdout; a
    21 KB line was cut in round 3) --; the full object goes to bench_detail.json (best effort: beside the script and under
    gpurun_out/) and to stderr."""
    full = json.dumps(line)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(full + "\n")
        except OSError:
            pass
    print(full, file=sys.stderr, flush=True)
    print(json.dumps(compact_line(line)), flush=True)


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


def pin_to_gpu_numa_node(torch, local_rank):
    """Keep this rank's host threads (and the page-locked buffers they allocate from now on) on the NUMA node its
    GPU hangs off -- the host-pointer pipeline moves 2.8 MB per frame over PCIe and spins in stream synchronisation.  Best effort:
    returns the node, or None where sysfs does not say (containers) or the affinity cannot be set."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:                                               # noqa: BLE001 -- never fail the bench over a placement hint
        return None


def rooflines(workload, w, h, d, B, stage_ms, stage_min, launches, fused):
    """roofline objects of the two heavy kernels of a leg from its HIP-event stage times (None where a stage did not run)."""
    counters = load_counters(workload)
    dp = -(-d // 16) * 16
    npaths = paths_of(workload)
    cells = w * h * d
    out = {"roofline": None, "roofline_sum_wta": None}
    # dominant kernel = the one-launch aggregation of all directions.  Bytes this dataflow has to move per frame: the
    # per-direction u8 L_r planes written once (1 B per cell of the padded volume and direction) + both census images and the
    # left image read once (9 B per pixel).  (Reference dataflow: 5 B per path evaluation)
    if stage_ms.get("aggregate"):
        out["roofline"] = kernel_roofline("sgm_aggregate_k", stage_ms["aggregate"], launches, stage_min["aggregate"], B, counters,
                                          w * h * dp * npaths + w * h * 9,
                                          f"W*H*Dp*{npaths} ({npaths} u8 L_r planes written once) + W*H*9 (census L/R + left image read once)",
                                          ref_equiv_bytes_per_frame=cells * 5 * npaths)
    if stage_ms.get("sum"):
        if fused:
            out["roofline_sum_wta"] = kernel_roofline("sgm_sum_wta_lr_k", stage_ms["sum"], launches, stage_min["sum"], B, counters,
                                                      w * h * dp * npaths + w * h * 8,
                                                      f"W*H*Dp*{npaths} ({npaths} planes read once) + W*H*8 (two disparity maps written)")
        else:                                                # D > 256: separate kernels, S written by the sum and re-read by the right view
            out["roofline_sum_wta"] = kernel_roofline("sgm_sum_wta_k", stage_ms["sum"], launches, stage_min["sum"], B, counters,
                                                      w * h * dp * (npaths + 2) + w * h * 4,
                                                      f"W*H*Dp*{npaths} (planes read once) + W*H*Dp*2 (S written) + W*H*4 (left map)")
            if stage_ms.get("wta"):
                out["roofline_wta_right"] = kernel_roofline("sgm_wta_right_k", stage_ms["wta"], launches, stage_min["wta"], B, counters,
                                                            w * h * dp * 2 + w * h * 4, "W*H*Dp*2 (S read once) + W*H*4 (right map written)")
    return out


def merge_timing(insts):
    """Per-kernel device time over EVERY match the instances timed since enable_timing (HIP events on the streams the kernels
    are launched on, one event set per match; at most 64 matches between two synchronisations are kept)."""
    stage_sum, stage_min, launches = {}, {}, 0
    for i in insts:
        mean, mn, cnt = i.mean_timing()
        launches += cnt
        for name in mean:
            stage_sum[name] = stage_sum.get(name, 0.0) + mean[name] * cnt
            stage_min[name] = min(stage_min.get(name, 1e30), mn[name])
    stage_ms = {k: v / launches for k, v in stage_sum.items()} if launches else {}
    return stage_ms, stage_min, launches


class HostPipeline:
    """The host-pointer boundary as a caller with a stream of frames uses it: n instances, each driven by its own host thread.  A step is
    one pass over a batch of n x B frames (at the defaults 4 x 8 = BASELINE config 4's batch of 32 KITTI frames): every instance
    takes a sub-batch of B frames through sgm_reset + sgm_match_async + sgm_match_wait on page-locked caller buffers (H2D, every
    kernel -- one launch per stage covers the B frames --, D2H).  Each instance writes its results into a small ring of output buffers, so the last
    `keep` batches of every instance can be verified after the timed region."""

    def __init__(self, S, device, w, h, opt, B, batches, n_inst, keep=4, cu_split="", overlap_post=True, honor4=False, timing=True):
        self.S, self.w, self.h, self.opt, self.B = S, w, h, opt, B
        self.insts = [S.SGMInstance(device, batch=B) for _ in range(n_inst)]
        self.keep = keep
        for i in self.insts:
            if honor4:
                i.set_honor_num_paths(True)
            if overlap_post and not i.set_overlap_post(True):     # a result is handed over by sgm_match_wait anyway
                raise SystemExit("sgm_set_overlap_post failed")
            if cu_split and not i.set_cu_split(cu_split):
                raise SystemExit(f"sgm_set_stage_cus failed for {cu_split!r}")
            if not i.reset(w, h, opt):
                raise SystemExit("sgm_reset failed")
        # page-locked input batches (shared, read-only) and per-instance output rings
        a = self.insts[0]
        self.inputs = []
        for ps in batches:
            L, R = a.host_array((B, h, w), np.uint8), a.host_array((B, h, w), np.uint8)
            for j, (l, r) in enumerate(ps):
                L[j], R[j] = l, r
            self.inputs.append((L, R))
        self.outs = [[i.host_array((B, h, w), np.float32) for _ in range(keep)] for i in self.insts]
        self.held = [[None] * keep for _ in self.insts]          # which input batch each output buffer holds
        self.done = [0] * n_inst
        if timing:
            for i in self.insts:
                i.enable_timing(True)

    def _worker(self, k, first_step, n_steps, barrier, fail):
        inst = self.insts[k]
        n = len(self.insts)
        barrier.wait()
        for s in range(first_step, first_step + n_steps):
            b = (s * n + k) % len(self.inputs)                   # step s = sub-batches s*n .. s*n + n-1 of the pool, one per instance
            slot = self.done[k] % self.keep
            L, R = self.inputs[b]
            if not (inst.reset(self.w, self.h, self.opt) and inst.match_async(L, R, self.outs[k][slot]) and inst.match_wait()):
                fail.append(k)
                return
            self.held[k][slot] = b
            self.done[k] += 1

    def run(self, first_step, n_steps, before=None):
        """Steps first_step .. first_step + n_steps - 1.  A step is one pass over n x B frames: every instance takes one sub-batch of B
        frames (one launch per stage covers them).  Returns the wall time from the moment every thread is ready to the moment the
        last result has been handed over."""
        n = len(self.insts)
        fail = []
        barrier = threading.Barrier(n + 1)
        th = [threading.Thread(target=self._worker, args=(k, first_step, n_steps, barrier, fail)) for k in range(n)]
        for t in th:
            t.start()
        if before:
            before()
        barrier.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        if fail:
            raise RuntimeError(f"a host-pointer match failed on instance(s) {sorted(set(fail))}")
        return el

    def run_for(self, seconds, first_step=0):
        """The same loop until `seconds` have passed; returns (elapsed, sub-batches of B frames processed)."""
        n = len(self.insts)
        fail, counts = [], [0] * n
        barrier = threading.Barrier(n + 1)
        stop = [0.0]

        def worker(k):
            inst = self.insts[k]
            barrier.wait()
            s = first_step
            while time.perf_counter() < stop[0]:
                b = (s * n + k) % len(self.inputs)
                slot = self.done[k] % self.keep
                L, R = self.inputs[b]
                if not (inst.reset(self.w, self.h, self.opt) and inst.match_async(L, R, self.outs[k][slot]) and inst.match_wait()):
                    fail.append(k)
                    return
                self.held[k][slot] = b
                self.done[k] += 1
                counts[k] += 1
                s += 1
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
        for t in th:
            t.start()
        stop[0] = time.perf_counter() + seconds + 0.01
        barrier.wait()
        t0 = time.perf_counter()
        stop[0] = t0 + seconds
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        if fail:
            raise RuntimeError(f"a host-pointer match failed on instance(s) {sorted(set(fail))}")
        return el, sum(counts)

    def verify(self, batch_seeds, digests):
        """(ok, bad, unpinned) over every frame the output rings hold."""
        ok = bad = unp = 0
        for k in range(len(self.insts)):
            for slot in range(self.keep):
                b = self.held[k][slot]
                if b is None:
                    continue
                for j in range(self.B):
                    sd = batch_seeds[b][j]
                    if sd not in digests:
                        unp += 1
                    elif digest(self.outs[k][slot][j]) == digests[sd]:
                        ok += 1
                    else:
                        bad += 1
        return ok, bad, unp

    def close(self):
        for i in self.insts:
            i.close()


def device_resident_leg(S, torch, device, workload, B, n_inst, steps, warmup, overlap_post=False, cu_split="", honor4=False, alone=False):
    """sgm_reset + sgm_match_device on frames that are already in HBM (kernels only: no PCIe), n_inst batches in flight on their
    own instances, one host thread; every frame of the last batch of each instance verified."""
    w, h, d, seed = WORKLOADS[workload]
    npaths = paths_of(workload)
    opt = S.default_option(d, num_paths=npaths, is_remove_speckles=not workload.endswith(SPECKLE_OFF))
    insts = [S.SGMInstance(device, batch=B) for _ in range(n_inst)]
    for i in insts:
        if honor4:
            i.set_honor_num_paths(True)
        if overlap_post and not i.set_overlap_post(True):
            raise SystemExit("sgm_set_overlap_post failed")
        if cu_split and not i.set_cu_split(cu_split):
            raise SystemExit(f"sgm_set_stage_cus failed for {cu_split!r}")
        if not i.reset(w, h, opt):
            raise SystemExit(f"sgm_reset failed for {workload}")
    digests = golden_digests(workload)
    n_batches = 2 if (not digests or (seed + 2 * B - 1) in digests) else 1     # two distinct batches where reference digests exist for both
    frames, seeds = [], []
    for k in range(n_batches):
        ps = [S.synth_pair(w, h, d, seed + k * B + j) for j in range(B)]
        seeds.append([seed + k * B + j for j in range(B)])
        frames.append((torch.from_numpy(np.stack([p[0] for p in ps])).cuda(), torch.from_numpy(np.stack([p[1] for p in ps])).cuda()))
    outs = [torch.empty((B, h, w), dtype=torch.float32, device="cuda") for _ in range(n_inst)]
    last = [None] * n_inst
    torch.cuda.synchronize()

    def step(k):
        i = insts[k % n_inst]
        l, r = frames[k % n_batches]
        if not (i.reset(w, h, opt) and i.match_device(l.data_ptr(), r.data_ptr(), outs[k % n_inst].data_ptr())):
            raise RuntimeError("sgm_match_device failed")
        last[k % n_inst] = k % n_batches

    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    for i in insts:
        i.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for i in insts:
        i.synchronize()
    stage_ms, stage_min, launches = merge_timing(insts)
    ok = bad = unp = 0
    for k in range(n_inst):
        if last[k] is None:
            continue
        got = outs[k].cpu().numpy()
        for j in range(B):
            sd = seeds[last[k]][j]
            if sd not in digests:
                unp += 1
            elif digest(got[j]) == digests[sd]:
                ok += 1
            else:
                bad += 1
    alone_ms = {}
    if alone and n_inst > 1:
        for rep in range(2):
            insts[0].enable_timing(True)
            for k in range(10):
                l, r = frames[k % n_batches]
                if not (insts[0].reset(w, h, opt) and insts[0].match_device(l.data_ptr(), r.data_ptr(), outs[0].data_ptr())):
                    raise RuntimeError("sgm_match_device failed")
            insts[0].synchronize()
        alone_ms = insts[0].mean_timing()[0]
    for i in insts:
        i.close()
    del frames, outs
    torch.cuda.empty_cache()
    n_fr = steps * B
    res = {"workload": workload, "width": w, "height": h, "disparity_range": d, "paths": npaths, "speckle_removal": bool(opt.is_remove_speckles),
           "entry": "sgm_reset + sgm_match_device (frames resident in HBM: kernels only)",
           "fps": round(n_fr / elapsed, 2), "value": round(w * h * d * npaths * n_fr / elapsed / 1e6, 1), "unit": "Mdisp/s",
           "ms_per_frame": round(elapsed / n_fr * 1e3, 4), "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps,
           "frames_per_step": B, "batches_in_flight": n_inst, "post_pass_on_second_stream": bool(overlap_post) or "post" in cu_split,
           "frames_verified": ok, "frames_mismatched": bad, "frames_without_reference_digest": unp,
           "stage_ms_per_batch_launch": {k: round(v, 4) for k, v in stage_ms.items()}}
    dp = -(-d // 16) * 16
    res.update(rooflines(workload, w, h, d, B, stage_ms, stage_min, launches, fused=dp <= 256))
    if alone_ms:
        for key, stage in (("roofline", "aggregate"), ("roofline_sum_wta", "sum")):
            rf = res.get(key)
            if rf and alone_ms.get(stage):
                ta = alone_ms[stage] * 1e-3
                al = {"avg_launch_ms": round(alone_ms[stage], 4), "batches_in_flight": 1}
                if rf.get("traffic"):
                    al["frac"] = round(rf["traffic"] / ta / 1e9 / HBM_PEAK_GBS, 4)
                al["algorithmic_frac"] = round(rf["algorithmic_bytes_per_launch"] / ta / 1e9 / HBM_PEAK_GBS, 4)
                if rf.get("valu"):
                    al["valu_frac"] = round(rf["valu"]["wave_insts_per_launch"] * VALU_ISSUE_CYCLES / (N_SIMD * CLOCK_HZ * ta), 4)
                rf["alone"] = al
    return res


def single_frame_leg(S, torch, device, workload):
    """Latency of ONE frame: a batch-1 instance, nothing else on the GPU (device buffers, no timing events in the timed calls)."""
    w, h, d, seed = WORKLOADS[workload]
    opt = S.default_option(d)
    l, r = S.synth_pair(w, h, d, seed)
    dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    solo = S.SGMInstance(device)
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    lat = []
    for it in range(12):
        if it == 8:
            solo.enable_timing(True)                 # the last 4 frames give the stage breakdown (events cost ~5 us each)
        solo.reset(w, h, opt)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        solo.match_device(dl.data_ptr(), dr.data_ptr(), out.data_ptr())
        solo.synchronize()
        if 2 <= it < 8:
            lat.append(time.perf_counter() - t1)
    stage = solo.mean_timing()[0]
    digests = golden_digests(workload)
    ok = (digest(out.cpu().numpy()) == digests[seed]) if seed in digests else None
    solo.close()
    return {"single_frame_latency_ms": round(float(np.median(lat)) * 1e3, 4), "stage_ms_single_frame": {k: round(v, 4) for k, v in stage.items()},
            "single_frame_verified": ok}


def blocking_leg(S, w, h, d, opt, pairs, seeds, digests, budget_s=2.0):
    """The reference contract as it stands: sgm_compute (SGM_Reset + SGM_Match) per frame on pageable arrays, one frame at a time."""
    g = S.SGM()
    n, t_sum, out = 0, 0.0, None
    buf = np.zeros((h, w), np.float32)                   # the caller's output buffer, allocated once as in the reference's main.c:86
    t_begin = time.perf_counter()
    while n < 400 and time.perf_counter() - t_begin < budget_s:
        l, r = pairs[n % len(pairs)]
        t0 = time.perf_counter()
        out = g.compute(l, r, opt, out=buf)
        dt = time.perf_counter() - t0
        if out is None:
            g.shutdown()
            return {"error": "sgm_compute failed"}
        if n >= 2:
            t_sum += dt
        n += 1
    g.shutdown()
    if n <= 2:
        return {"error": "too few frames"}
    ms = t_sum / (n - 2) * 1e3
    sd = seeds[(n - 1) % len(pairs)]
    return {"ms_per_frame": round(ms, 4), "fps": round(1e3 / ms, 1), "frames": n - 2,
            "entry": "sgm_compute (SGM_Reset + SGM_Match), pageable numpy arrays, one frame per call",
            "verified": (digest(out) == digests[sd]) if sd in digests else None}


def blocking_pinned_leg(S, device, w, h, opt, pairs, seeds, digests, budget_s=1.0):
    """One frame at a time as above, but the caller keeps its images and its result in page-locked memory (sgm_host_alloc): no staging
    copies, the DMA engines work on the caller's buffers.  The frame is put into the page-locked arrays outside the timed call (a
    camera driver would have captured it there)."""
    i = S.SGMInstance(device)
    try:
        if not i.reset(w, h, opt):
            return {"error": "sgm_reset failed"}
        L, R, O = i.host_array((h, w), np.uint8), i.host_array((h, w), np.uint8), i.host_array((h, w), np.float32)
        n, t_sum = 0, 0.0
        t_begin = time.perf_counter()
        while n < 400 and time.perf_counter() - t_begin < budget_s:
            L[...], R[...] = pairs[n % len(pairs)]
            t0 = time.perf_counter()
            ok = i.reset(w, h, opt) and i.match_async(L, R, O) and i.match_wait()
            dt = time.perf_counter() - t0
            if not ok:
                return {"error": "sgm_match_async / sgm_match_wait failed"}
            if n >= 2:
                t_sum += dt
            n += 1
        if n <= 2:
            return {"error": "too few frames"}
        ms = t_sum / (n - 2) * 1e3
        sd = seeds[(n - 1) % len(pairs)]
        return {"ms_per_frame": round(ms, 4), "fps": round(1e3 / ms, 1), "frames": n - 2,
                "entry": "sgm_reset + sgm_match_async + sgm_match_wait, one frame per call, caller buffers from sgm_host_alloc",
                "verified": (digest(O) == digests[sd]) if sd in digests else None}
    finally:
        i.close()


def stream_leg(S, device, workload, n_frames, B, n_inst):
    """BASELINE config 5 as a stream: n_frames DISTINCT frames (seed = first + f) through the host-pointer pipeline, every input
    and output in its own page-locked buffer; sustained rate over the whole stream; the first and the last four frames verified."""
    from concurrent.futures import ThreadPoolExecutor
    w, h, d, seed = WORKLOADS[workload]
    opt = S.default_option(d)
    insts = [S.SGMInstance(device, batch=B) for _ in range(n_inst)]
    for i in insts:
        if not (i.set_overlap_post(True) and i.reset(w, h, opt)):
            raise SystemExit("stream leg: instance set-up failed")
    n_batches = n_frames // B
    a = insts[0]
    L = [a.host_array((B, h, w), np.uint8) for _ in range(n_batches)]
    R = [a.host_array((B, h, w), np.uint8) for _ in range(n_batches)]
    O = [a.host_array((B, h, w), np.float32) for _ in range(n_batches)]
    lib = S.load_library()

    def synth(f):
        lib.SGM_SynthPair(w, h, d, (seed + f) & 0xFFFFFFFF, L[f // B][f % B].ctypes.data, R[f // B][f % B].ctypes.data)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(synth, range(n_batches * B)))
    t_synth = time.perf_counter() - t0
    fail = []
    barrier = threading.Barrier(n_inst + 1)

    def worker(k):
        inst = insts[k]
        # one warm-up batch (its result is overwritten by the timed pass)
        ok = inst.reset(w, h, opt) and inst.match_async(L[k], R[k], O[k]) and inst.match_wait()
        barrier.wait()
        for b in range(k, n_batches, n_inst):
            if not (ok and inst.reset(w, h, opt) and inst.match_async(L[b], R[b], O[b]) and inst.match_wait()):
                fail.append(k)
                return
    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_inst)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    digests = golden_digests(workload)
    ok = bad = 0
    checked = [f for f in list(range(4)) + list(range(n_frames - 4, n_frames)) if (seed + f) in digests and f < n_batches * B]
    for f in checked:
        if digest(O[f // B][f % B]) == digests[seed + f]:
            ok += 1
        else:
            bad += 1
    for i in insts:
        i.close()
    frames = n_batches * B
    return {"workload": workload, "width": w, "height": h, "disparity_range": d, "paths": 8, "frames": frames, "distinct_frames": frames,
            "entry": "sgm_reset + sgm_match_async + sgm_match_wait, page-locked buffers, one host thread per instance",
            "frames_per_call": B, "instances": n_inst, "seconds": round(el, 4), "fps": round(frames / el, 2),
            "value": round(w * h * d * 8 * frames / el / 1e6, 1), "unit": "Mdisp/s", "failed": bool(fail),
            "frames_checked": [int(f) for f in checked], "frames_verified": ok, "frames_mismatched": bad,
            "synthesis_seconds_not_timed": round(t_synth, 2)}


def run_frames(args):
    import torch
    import torch.distributed as dist
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd.sharding import frames_of_rank

    world, rank, local_rank, backend = init_dist(args)
    numa_node = pin_to_gpu_numa_node(torch, local_rank) if int(os.environ.get("SGM_BENCH_PIN", "1")) else None
    w, h, d, seed = WORKLOADS[args.workload]
    npaths = paths_of(args.workload)
    honor4 = npaths == 4
    opt = S.default_option(d, num_paths=npaths, is_remove_speckles=not args.workload.endswith(SPECKLE_OFF))
    B = max(1, args.batch)
    n_host = max(1, args.host_instances)
    digests = golden_digests(args.workload)
    legs = set(args.legs.split(",")) if args.legs else {"headline", "sustained", "device", "latency", "host", "workloads", "stream", "cpu"}
    if world > 1:
        legs &= {"headline", "device"}
    if args.no_host_boundary:
        legs -= {"host"}
    if args.no_cpu_baseline:
        legs -= {"cpu"}

    # ---- this rank's frames: its share of the pool (config 4: a batch of 32 frames sharded frame by frame), cut into batches
    #      of B; a rank with fewer than 2 B frames cycles through its share
    pool = POOL_FRAMES if len(digests) >= POOL_FRAMES else max(2 * B, 1)
    mine = frames_of_rank(pool, world, rank) or [rank % max(pool, 1)]
    n_batches = max(n_host, 2, len(mine) // B)
    batch_frames = [[mine[(k * B + j) % len(mine)] for j in range(B)] for k in range(n_batches)]
    batch_seeds = [[seed + f for f in fr] for fr in batch_frames]
    pair_of = {f: S.synth_pair(w, h, d, seed + f) for f in sorted({f for fr in batch_frames for f in fr})}
    batches = [[pair_of[f] for f in fr] for fr in batch_frames]

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- headline: the host-pointer boundary (SURVEY.md 8d's t_frame) -------------------------------------------------------
    overlap_post = args.overlap_post if args.overlap_post is not None else int(os.environ.get("SGM_BENCH_OVERLAP_POST", "1"))
    cu_split = args.cu_split if args.cu_split is not None else os.environ.get("SGM_BENCH_CU_SPLIT", "")
    hp = HostPipeline(S, local_rank, w, h, opt, B, batches, n_host, keep=int(os.environ.get("SGM_BENCH_KEEP", "4")), cu_split=cu_split,
                      overlap_post=bool(overlap_post), honor4=honor4, timing=bool(int(os.environ.get("SGM_BENCH_TIMING", "1"))))
    hp.run(0, args.warmup)
    torch.cuda.synchronize()
    if int(os.environ.get("SGM_BENCH_TIMING", "1")):
        for i in hp.insts:
            i.enable_timing(True)                        # new statistics window: only the timed region is averaged

    def before_timed():
        barrier()
        torch.cuda.synchronize()
    elapsed = hp.run(args.warmup, args.steps, before=before_timed)
    torch.cuda.synchronize()
    barrier()
    stage_ms, stage_min, launches = merge_timing(hp.insts)
    n_ok, n_bad, n_unp = hp.verify(batch_seeds, digests)
    sustained = None
    if "sustained" in legs:
        el2, steps2 = hp.run_for(max(2.0, args.sustain_seconds), first_step=args.warmup + args.steps)
        ok2, bad2, unp2 = hp.verify(batch_seeds, digests)
        sustained = {"seconds": round(el2, 3), "sub_batches": steps2, "frames": steps2 * B, "fps": round(steps2 * B / el2, 2),
                     "value": round(w * h * d * npaths * steps2 * B / el2 / 1e6, 1), "unit": "Mdisp/s",
                     "frames_verified": ok2, "frames_mismatched": bad2,
                     "note": "the headline loop run for a fixed time instead of a fixed number of steps"}
    hp.close()
    if world > 1:
        t = torch.tensor([elapsed, float(n_ok), float(n_bad), float(n_unp)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        n_ok, n_bad, n_unp = int(t[1].item()), int(t[2].item()), int(t[3].item())

    # ---- the same frames already resident in HBM (kernels only), this rank ------------------------------------------------
    dev = None
    if "device" in legs:
        dev = device_resident_leg(S, torch, local_rank, args.workload, B, max(1, args.in_flight), max(10, min(args.steps, 80)),
                                  min(args.warmup, 10), overlap_post=False, cu_split="", honor4=honor4, alone=bool(args.alone))
        if world > 1:
            t = torch.tensor([dev["fps"]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dev["fps_all_ranks"] = round(float(t[0].item()), 2)

    if rank == 0:
        total_frames = args.steps * B * n_host * world
        cells = w * h * d
        value = cells * npaths * total_frames / elapsed / 1e6
        line = {
            "metric": "Mdisp/s (W*H*D*paths per second), fps beside it; t_frame includes H2D of both images and D2H of the disparity map",
            "value": round(value, 1), "unit": "Mdisp/s",
            "fps": round(total_frames / elapsed, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u16 integer min-plus (f32 sub-pixel tail)", "data": "synthetic",
            "config": {"workload": args.workload, "mode": "frames", "width": w, "height": h, "disparity_range": d, "paths": npaths,
                       "stages": f"census+cost+aggregate{npaths}+wta+lrcheck+" + ("speckle+" if opt.is_remove_speckles else "") + "median",
                       "entry": "sgm_reset + sgm_match_async + sgm_match_wait on page-locked host buffers: H2D + kernels + D2H per step "
                                "(the reference's SGM_Reset + SGM_Match contract, SemiGlobalMatching.c:77-78,122)",
                       "frames_per_step": B * n_host, "frames_per_launch": B, "frames_per_gpu": args.steps * B * n_host,
                       "step": f"one pass over a batch of {B * n_host} frames: {n_host} instances x one sub-batch of {B} frames each (every kernel covers its {B} frames in one launch)",
                       "instances_per_gpu": n_host, "host_threads_per_gpu": n_host,
                       "post_pass_on_second_stream": bool(overlap_post) or "post" in cu_split, "stage_cus_per_xcd": cu_split or None,
                       "frame_pool": pool, "frames_of_rank0": mine,
                       "sharding": "independent frames per rank (sharding.frames_of_rank over the pool), no collective",
                       "rank0_host_threads_pinned_to_numa_node": numa_node},
            "ms_per_frame": round(elapsed / (args.steps * B * n_host) * 1e3, 4),
            "stage_ms_per_batch_launch": {k: round(v, 4) for k, v in stage_ms.items()},
            "frames_verified": n_ok, "frames_mismatched": n_bad, "frames_without_reference_digest": n_unp,
            "verified_against_golden": (n_bad == 0 and n_ok > 0 and n_unp == 0),
            "verification": "sha256 of every frame of the last 4 batches of each instance (all ranks), copied back by the timed calls "
                            "themselves, vs the reference's own C for the same seeds (tests/golden/bench_frames.json)",
            "source_id": source_id(),
        }
        dp = -(-d // 16) * 16
        line.update(rooflines(args.workload, w, h, d, B, stage_ms, stage_min, launches, fused=dp <= 256))
        for key in ("roofline", "roofline_sum_wta"):
            if line.get(key):
                line[key]["timed_over"] = "every launch of the timed region (host-pointer pipeline), HIP events on the launching stream"
        counters = load_counters(args.workload)
        if counters and counters.get("kernels"):
            tot = sum(v.get("hbm_bytes_per_frame") or 0 for v in counters["kernels"].values())
            if tot:
                fr_s = total_frames / world / elapsed
                line["frame_traffic"] = {"hbm_bytes_per_frame": int(tot), "GBps_per_gpu": round(tot * fr_s / 1e9, 1),
                                         "frac_of_peak": round(tot * fr_s / 1e9 / HBM_PEAK_GBS, 4),
                                         "stale": counters.get("source_id") != source_id()}
        line["frame_reference_dataflow_equiv"] = {"bytes_per_frame": cells * (5 * npaths + 3),
                                                  "GBps_per_gpu": round(cells * (5 * npaths + 3) * total_frames / world / elapsed / 1e9, 1),
                                                  "note": "W*H*D*(5*paths+3) B of the reference dataflow (SURVEY.md 8d) per frame time; not a roofline fraction"}
        if sustained:
            line["sustained"] = sustained
        if dev:
            line["device_resident"] = dev
            # the kernels' own quality (one batch in flight, nothing else on the GPU) beside the timed configuration's launch times
            for key in ("roofline", "roofline_sum_wta"):
                if line.get(key) and dev.get(key) and dev[key].get("alone"):
                    line[key]["alone"] = dict(dev[key]["alone"], measured_in="device_resident leg: the same kernel, 8 frames per launch, one "
                                                                              "instance, nothing else on the GPU")
        if world == 1:
            if "latency" in legs:
                line.update(single_frame_leg(S, torch, local_rank, args.workload))
            if "host" in legs:
                pairs = [pair_of[f] for f in sorted(pair_of)][:16]
                sds = [seed + f for f in sorted(pair_of)][:16]
                hb = {"blocking_single_frame": blocking_leg(S, w, h, d, opt, pairs, sds, digests),
                      "blocking_single_frame_pinned": blocking_pinned_leg(S, local_rank, w, h, opt, pairs, sds, digests)}
                # the same pipeline with PAGEABLE caller buffers (staged through the instances' own pinned buffers)
                pg = [S.SGMInstance(local_rank, batch=B) for _ in range(n_host)]
                try:
                    bufs = []
                    for i in pg:
                        i.set_overlap_post(True)
                        assert i.reset(w, h, opt)
                        bufs.append((np.stack([p[0] for p in batches[0]]), np.stack([p[1] for p in batches[0]]), np.empty((B, h, w), np.float32)))
                    stop = [0.0]
                    rounds = [0] * len(pg)

                    def worker(k):
                        L, R, O = bufs[k]
                        while time.perf_counter() < stop[0]:
                            if not (pg[k].reset(w, h, opt) and pg[k].match_async(L, R, O) and pg[k].match_wait()):
                                return
                            rounds[k] += 1
                    for k in range(len(pg)):
                        assert pg[k].match_async(*bufs[k]) and pg[k].match_wait()
                    t0 = time.perf_counter()
                    stop[0] = t0 + 1.5
                    th = [threading.Thread(target=worker, args=(k,)) for k in range(len(pg))]
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    el = time.perf_counter() - t0
                    ver = all(digest(bufs[k][2][j]) == digests.get(batch_seeds[0][j]) for k in range(len(pg)) for j in range(B))
                    hb["pipelined_pageable"] = {"fps": round(sum(rounds) * B / el, 1), "frames": sum(rounds) * B, "instances": len(pg),
                                                "entry": "sgm_reset + sgm_match_async + sgm_match_wait, pageable caller buffers", "verified": ver}
                finally:
                    for i in pg:
                        i.close()
                for v in hb.values():
                    if "fps" in v:
                        v["vs_headline"] = round(v["fps"] / line["fps"], 3)
                line["host_boundary"] = hb
            if "workloads" in legs:
                wl = []
                for name, wb, wf, ws in (("cone_450x375_d64_p8", 8, 2, 40), ("cone_450x375_d64_p4", 8, 2, 40),
                                         ("middlebury_2880x1988_d256_p8", 2, 2, 8), ("drivingstereo_1762x800_d192_p8", 8, 2, 8),
                                         ("kitti_1242x375_d128_p8_nospeckle", 8, 2, 20)):
                    try:
                        wl.append(device_resident_leg(S, torch, local_rank, name, wb, wf, ws, 3, honor4=name.endswith("_p4"), alone=True))
                    except Exception as e:                       # one workload must not take the line down
                        wl.append({"workload": name, "error": repr(e)})
                line["workloads"] = wl
            if "stream" in legs:
                try:
                    line["stream"] = stream_leg(S, local_rank, "drivingstereo_1762x800_d192_p8", args.stream_frames, 8, 3)
                except Exception as e:
                    line["stream"] = {"error": repr(e)}
            if "cpu" in legs and not args.no_cpu_baseline:
                cb = cpu_baseline(w, h, d, seed)
                try:
                    cb["cores_16"] = cpu_baseline_all_cores(w, h, d, seed)
                except Exception as e:                                   # the one-core figure is the contract; this one is extra
                    cb["cores_16"] = {"error": repr(e)}
                line["cpu_baseline"] = cb
                line["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
            else:
                line["cpu_baseline"] = None
        else:
            line["cpu_baseline"] = None
        emit(line)

    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps (a step = one batch of host-instances x batch frames through the host-pointer boundary)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="frames", choices=["frames", "tiles"])
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per step, one launch per stage covers them all (frames mode: default 8; tiles mode: default 1 = "
                         "one frame cut into the ranks' tiles per step, N = the same tile of N frames per launch)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--in-flight", type=int, default=None,
                    help="frames mode: instances (HIP streams) the device-resident leg round-robins batches over (default 2); "
                         "tiles mode: frames in flight through the rank pipeline (default 2 x ranks)")
    ap.add_argument("--host-instances", type=int, default=int(os.environ.get("SGM_BENCH_HOST_INSTANCES", "4")),
                    help="frames mode: instances (one host thread each) the headline's host-pointer pipeline round-robins steps over")
    ap.add_argument("--legs", default=None,
                    help="comma list of the legs to run (default all at --gpus 1): headline is always run; sustained, device, latency, "
                         "host, workloads, stream, cpu")
    ap.add_argument("--sustain-seconds", type=float, default=2.0)
    ap.add_argument("--stream-frames", type=int, default=256)
    ap.add_argument("--tile-ranks-in-process", type=int, default=0,
                    help="tiles mode on ONE GPU: this many tile ranks as threads of one process (device copies stand in for xGMI): "
                         "what the pipeline schedule itself costs against --mode tiles with one rank; not a multi-GPU measurement")
    ap.add_argument("--tile-lead", type=int, default=int(os.environ.get("SGM_TILE_LEAD", "2")),
                    help="tiles mode: steps tile_begin of a frame is queued ahead of its first sweep (tiling.TilePipeline lead)")
    ap.add_argument("--tile-host", default=os.environ.get("SGM_TILE_HOST", "c"), choices=["c", "python"],
                    help="tiles mode: c = the library's own pipeline (include/sgm_tiles.h: schedule, slots, streams, events and the RCCL "
                         "transport in C; default), python = tiling.TilePipeline driving the same C schedule with a Python engine over "
                         "torch.distributed (what a gloo rehearsal on a one-GPU box needs)")
    ap.add_argument("--tile-rank-alone", default=None, metavar="r/N",
                    help="tiles mode: rank r of an N-rank pipeline alone on this GPU with the exchanges skipped: its time per frame "
                         "(a projection input for N GPUs, results are not produced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true")
    ap.add_argument("--overlap-post", type=int, default=None, choices=[0, 1],
                    help="sgm_set_overlap_post on the headline's instances: LR check / speckle / median of a batch on a second stream "
                         "beside the next batch's aggregation (default: SGM_BENCH_OVERLAP_POST or 1; the device-resident leg keeps one stream)")
    ap.add_argument("--cu-split", default=None, metavar="SPEC",
                    help="sgm_set_stage_cus on the bench's instances: stage groups on streams and compute units of their own, e.g. "
                         "'post=0:2,sum=2:8,main=10:22' = first:count CUs of every XCD (count 0: own stream on all CUs)")
    ap.add_argument("--alone", type=int, default=1, choices=[0, 1], nargs="?", const=1,
                    help="device-resident leg: after its timed region also time the batches with ONE instance and nothing else on the GPU -> "
                         "device_resident.roofline.alone (the kernel's own quality; in the timed regions other batches share the chip).  "
                         "On by default; the kernel-trace profile of tools/refresh_profiles.sh runs --legs headline, which has no such launches")
    args = ap.parse_args()
    if args.mode == "tiles":
        # a rank of the tile pipeline keeps ~N + 3 HIP streams busy, some with millisecond-long serial kernels (tile_begin's
        # horizontal lines); on HIP's default of 4 hardware queues short kernels queue up behind those (measured, NOTES.md
        # section 7: 1.9 -> 1.45 ms per frame for one rank of eight at 3840x2160).  Read by the runtime when it starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        import soc_project_stereo_matching_amd.tile_bench as tb
        args.workload = args.workload or "uhd_3840x2160_d128_p8"
        python_host = args.tile_host == "python" or os.environ.get("SGM_BENCH_BACKEND", "nccl") != "nccl"
        if args.tile_rank_alone:
            return (tb.run_tile_rank_alone if python_host else tb.run_tile_rank_alone_c)(args, args.tile_rank_alone, WORKLOADS)
        if args.tile_ranks_in_process > 1:
            return (tb.run_tiles_in_process if python_host else tb.run_tiles_in_process_c)(args, args.tile_ranks_in_process, WORKLOADS, golden_digests)
        return (tb.run_tiles if python_host else tb.run_tiles_c)(args, init_dist, WORKLOADS, golden_digests)
    if args.cu_split or os.environ.get("SGM_BENCH_CU_SPLIT"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # up to three streams per instance in flight: no two of them on one hardware queue
    args.workload = args.workload or "kitti_1242x375_d128_p8"
    args.batch = args.batch or 8
    args.in_flight = args.in_flight or 2
    run_frames(args)


if __name__ == "__main__":
    main()
