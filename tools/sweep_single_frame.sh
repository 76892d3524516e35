#!/usr/bin/env bash
# sweep_single_frame.sh -- single-frame stage times (tools/single_frame_timing.py, default layout only) for the in-tree library and
# every build under variants/ (tools/build_variant.sh).  On the GPU box.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for lib in intree variants/*/libsgm_mi355x.so; do
  if [ "$lib" = intree ]; then unset SGM_LIBRARY_PATH; name=intree; else export SGM_LIBRARY_PATH="$PWD/$lib"; name=$(basename "$(dirname "$lib")"); fi
  echo "== $name"
  SGM_SINGLE_ONLY_DEFAULT=1 timeout -k 10 120 python tools/single_frame_timing.py "$@" 2>/dev/null | head -1
done
