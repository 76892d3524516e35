#!/usr/bin/env bash
# bench_repeats.sh [N] -- the driver's command N times back to back on one box, one summary line per run.  On the GPU box.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
echo "run  headline_fps(20 steps)  sustained_fps  device_resident_fps  single_frame_ms  blocking_fps  blocking_pinned_fps  pageable_pipelined_fps  [cone8, cone4, 2880x1988x256, 1762x800x192, kitti_nospeckle]_fps  stream_fps  mismatched"
for i in $(seq 1 "${1:-3}"); do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); hb=d["host_boundary"]
bad=d["frames_mismatched"]+d["device_resident"]["frames_mismatched"]+sum(w.get("frames_mismatched",1) for w in d["workloads"])+d["stream"]["frames_mismatched"]
print(sys.argv[1], d["fps"], d["sustained"]["fps"], d["device_resident"]["fps"], d["single_frame_latency_ms"], hb["blocking_single_frame"]["fps"], hb["blocking_single_frame_pinned"]["fps"], hb["pipelined_pageable"]["fps"], [w.get("fps") for w in d["workloads"]], d["stream"]["fps"], bad)' "$i" || exit 1
done
