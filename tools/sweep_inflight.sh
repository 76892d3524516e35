#!/usr/bin/env bash
# usage: tools/sweep_inflight.sh "1 2 4 8" [extra bench args]   (GPU box)
for f in $1; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --in-flight $f --steps 300 --warmup 30 ${@:2} 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("in_flight", d["config"]["frames_in_flight_per_gpu"], "fps", d["fps"], "ms/step", d["ms_per_step"], "lat", d["single_frame_latency_ms"], d["stage_ms_in_flight"])'
done
