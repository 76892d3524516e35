#!/usr/bin/env bash
# Runs bench.py (KITTI batches, no CPU baseline / host boundary) once per library build under variants/ and
# once on the in-tree build, with one and two batches in flight; prints fps and the two heavy kernels' launch times.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for lib in intree variants/*/libsgm_mi355x.so; do
  for fl in 1 2; do
    if [ "$lib" = intree ]; then unset SGM_LIBRARY_PATH; name=intree; else export SGM_LIBRARY_PATH="$PWD/$lib"; name=$(basename "$(dirname "$lib")"); fi
    out=gpurun_out/sweep_${name}_f${fl}.json
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --in-flight $fl --no-cpu-baseline --no-host-boundary > "$out" 2>> gpurun_out/sweep.err || { echo "$name f$fl FAILED"; exit 1; }
    python - "$out" "$name" "$fl" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d["stage_ms_per_batch_launch"]
print(f"{sys.argv[2]:28s} in-flight {sys.argv[3]}: {d['fps']:8.1f} fps  aggregate {s['aggregate']:.4f}  sum {s['sum']:.4f}  median {s['median']:.4f}  verified {d['frames_verified']}/{d['frames_verified'] + d['frames_mismatched']}", flush=True)
PY
  done
done
