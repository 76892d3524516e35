#!/usr/bin/env bash
# build_variant.sh NAME [make variables...] -- builds the library into variants/NAME/libsgm_mi355x.so with other
# compiler settings (e.g. AGG_SCHED=iterative-ilp) without touching the in-tree build; run one with
#   SGM_LIBRARY_PATH=variants/NAME/libsgm_mi355x.so python bench.py ...
set -euo pipefail
name=$1; shift
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/variants/$name"
mkdir -p "$out/csrc"
cp "$root"/soc_project_stereo_matching_amd/csrc/*.{hip,hpp,h,c} "$root"/soc_project_stereo_matching_amd/csrc/Makefile "$out/csrc/"
mkdir -p "$out/include" && cp "$root"/include/*.h "$out/include/"
sed -i "s#\.\./\.\./include#../include#g" "$out"/csrc/*.c "$out"/csrc/*.h "$out/csrc/Makefile"
make -s -C "$out/csrc" -j4 "$@" ../libsgm_mi355x.so
echo "built $out/libsgm_mi355x.so ($*)"
