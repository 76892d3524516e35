#!/usr/bin/env python3
"""What the row-tile pipeline costs and what N GPUs could reach, measured on ONE GPU.

For every batch size: bench.py --mode tiles with one rank (the whole frame), with N ranks as threads of one process
(--tile-ranks-in-process: every hand-over and gather in place as device copies, digests verified), and one rank of N alone with
the exchanges skipped (--tile-rank-alone r/N: that rank's share of the launches, nothing else on the GPU).  1 / max over the
ranks of the last figure is the rate N GPUs reach if the exchanges hide completely: a projection, not a multi-GPU result.

    python tools/tiles_schedule_cost.py [workload] > gpurun_out/tiles_schedule_cost.json        (each run is a child process)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, key):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "tiles"] + args, capture_output=True, text=True, timeout=600)
    for line in out.stdout.splitlines():
        if line.startswith("{") and key in line:
            return json.loads(line)
    raise RuntimeError(f"bench.py {' '.join(args)} failed:\n{out.stderr[-2000:]}")


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "uhd_3840x2160_d128_p8"
    kitti = wl.startswith("kitti")
    res = {"workload": wl, "note": "one MI355X; GPU_MAX_HW_QUEUES=16 (bench.py's default for tiles mode)", "batches": []}
    for b in (1, 4, 8):
        steps = str((240 if kitti else 48) // b + 8)
        common = ["--workload", wl, "--batch", str(b), "--steps", steps, "--warmup", "4"]
        one = run(common + ["--tile-lead", "0", "--in-flight", "3"], '"metric"')   # one rank: nothing to lead; three slots of whole frames
        entry = {"frames_per_step": b,
                 "one_rank": {"fps": one["fps"], "frames_verified": one["frames_verified"], "frames_mismatched": one["frames_mismatched"]},
                 "ranks_in_process": [], "rank_alone": []}
        for n in (2, 4):
            try:
                d = run(common + ["--tile-ranks-in-process", str(n)], '"metric"')
                entry["ranks_in_process"].append({"ranks": n, "fps": d["fps"], "frames_verified": d["frames_verified"],
                                                  "frames_mismatched": d["frames_mismatched"]})
            except RuntimeError as exc:          # N ranks x (N + 2) slots x B frames of planes on ONE GPU: 3840x2160 x 8 does not fit
                entry["ranks_in_process"].append({"ranks": n, "error": "out of memory" if "out of memory" in str(exc) else str(exc)[-300:]})
        for n in (2, 4, 8):
            worst = 0.0
            for r in sorted({0, n // 2, n - 1}):
                d = run(common + ["--tile-rank-alone", f"{r}/{n}"], "tile_rank_alone")
                entry["rank_alone"].append({"rank": f"{r}/{n}", "rows": d["rows"], "ms_per_frame": d["ms_per_frame"]})
                worst = max(worst, d["ms_per_frame"])
            entry.setdefault("projection", []).append({"gpus": n, "fps_if_exchanges_hide": round(1e3 / worst, 1),
                                                       "x_one_rank": round(1e3 / worst / one["fps"], 2)})
        res["batches"].append(entry)
        print(f"batch {b}: one rank {one['fps']} fps; projection {entry['projection']}", file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
