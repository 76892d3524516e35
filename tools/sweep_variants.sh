#!/usr/bin/env bash
# Runs bench.py (KITTI batches; device-resident leg alone and in flight, sustained host pipeline) once per library build under
# variants/ (tools/build_variant.sh) and once on the in-tree build.  On the GPU box.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for lib in intree variants/*/libsgm_mi355x.so; do
    if [ "$lib" = intree ]; then unset SGM_LIBRARY_PATH; name=intree; else export SGM_LIBRARY_PATH="$PWD/$lib"; name=$(basename "$(dirname "$lib")"); fi
    out=gpurun_out/sweep_${name}.json
    timeout -k 10 300 python bench.py --steps 40 --warmup 8 --legs device,sustained --alone "$@" > "$out" 2>> gpurun_out/sweep.err || { echo "$name FAILED"; continue; }
    python - "$out" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
dv = d["device_resident"]; s = dv["stage_ms_per_batch_launch"]; al = dv["roofline"].get("alone", {}); als = dv["roofline_sum_wta"].get("alone", {})
print(f"{sys.argv[2]:16s} host {d['sustained']['fps']:7.1f} fps  device {dv['fps']:7.1f} fps  agg {s['aggregate']:.3f} (alone {al.get('avg_launch_ms')})  sum {s['sum']:.3f} (alone {als.get('avg_launch_ms')})  verified {dv['frames_verified']}/{dv['frames_verified'] + dv['frames_mismatched']}", flush=True)
PY
done
