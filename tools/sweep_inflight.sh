#!/usr/bin/env bash
# usage: tools/sweep_inflight.sh "1 2 4 8" [extra bench args]   (GPU box)
for f in $1; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --in-flight $f ${@:2} 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("in_flight", d["config"]["batches_in_flight_per_gpu"], "batch", d["config"]["frames_per_step"], "fps", d["fps"], "ms/step", d["ms_per_step"], "lat", d["single_frame_latency_ms"], "ms/frame", d["ms_per_frame"], d["stage_ms_per_batch_launch"])'
done
