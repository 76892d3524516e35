#!/usr/bin/env bash
# build_ref.sh W H D -- TEST INFRASTRUCTURE ONLY.
#
# Builds the REFERENCE's own SemiGlobalMatching.c (read in place from /root/reference; nothing is
# copied into the repo, the patched text only ever exists in a pipe) into
#   oracle/_ref/libsgm_ref_<W>x<H>x<D>.so      (git-ignored; travels to the GPU box as a binary)
# with capacity W x H x D and the two edits SURVEY.md 8(c) defines for "the oracle build":
#   1. guard: a diagonal path step whose pixel pointer has left the image ends the line
#      (inserted in front of SemiGlobalMatching.c:325; the stock code reads/writes out of bounds
#      there = undefined behaviour, Q6);
#   2. the uint8_t loop counter at SemiGlobalMatching.c:272 is widened so D >= 256 terminates (Q2).
# Each edit must match exactly one source line or the build fails.
set -euo pipefail
W=${1:?width} H=${2:?height} D=${3:?disparity range}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${SGM_REFERENCE_DIR:-/root/reference}/SemiGlobalMatching/SemiGlobalMatching"
SRC="$REF/SemiGlobalMatching.c"
[ -f "$SRC" ] || { echo "build_ref.sh: reference not present at $SRC (nothing built)"; exit 3; }
OUT="$HERE/_ref"
mkdir -p "$OUT"

# the reference's buffers are static arrays: past 2 GB of .bss the default code model cannot address them
MODEL=""
[ $((W * H * D)) -gt 300000000 ] && MODEL="-mcmodel=medium"

GUARD_RE='^[[:space:]]*gray = \*img_pos;'
COUNTER_RE='for (uint8_t f = 0; f < sgm\.disp_range; f++)'
[ "$(grep -c "$GUARD_RE" "$SRC")" = 1 ]   || { echo "guard anchor not unique"; exit 4; }
[ "$(grep -c "$COUNTER_RE" "$SRC")" = 1 ] || { echo "counter anchor not unique"; exit 4; }

{
  cat "$HERE/ref_prelude.h"
  sed -e "/$GUARD_RE/i if (img_pos < img_data || img_pos >= img_data + (size_t)sgm.width * sgm.height) { ref_oob_dropped++; break; }" \
      -e "s/$COUNTER_RE/for (uint16_t f = 0; f < sgm.disp_range; f++)/" "$SRC"
  cat "$HERE/ref_harness_tail.c"
} | gcc -O2 -w $MODEL -x c - -I"$REF" -DREF_W="$W" -DREF_H="$H" -DREF_D="$D" \
        -shared -fPIC -o "$OUT/libsgm_ref_${W}x${H}x${D}.so" -lm
echo "built $OUT/libsgm_ref_${W}x${H}x${D}.so"
