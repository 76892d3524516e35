#!/usr/bin/env python3
"""A call SEQUENCE through the reference's own entry points, shapes changing between frames -- produced by the REFERENCE ITSELF.

The reference's census buffers are static arrays that SGM_Reset never clears and census_transform_5x5 writes in the
interior only (SemiGlobalMatching.h:67-68, .c:136,140-141): after a Reset to another shape the 2-pixel
border -- every pixel for W <= 5 or H <= 5 -- keeps what earlier frames left at the same LINEAR index, and the result of
a frame depends on the frames before it.  This script records, for one fixed sequence run in one process on zeroed
statics, the digest of every frame's output (SGM_Reset + SGM_Match, nothing cleared in between) and, beside it, the
digest of the same frame computed on zeroed statics ("fresh": what a new process would return).

    oracle/build_ref.sh 450 375 64 && python tests/golden/make_golden_history.py   -> tests/golden/census_history.json

Expected values come from oracle/_ref/libsgm_ref_450x375x64.so (the reference's SemiGlobalMatching.c, guarded build)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "census_history.json")

# (W, H, dmin, dmax, seed): larger -> smaller -> the census no-op sizes (W <= 5 / H <= 5: every word stale) -> larger again
SEQUENCE = [(200, 120, 0, 32, 0xC0DE01), (90, 60, 0, 16, 0xC0DE02), (5, 30, 0, 4, 0xC0DE03), (64, 40, 2, 18, 0xC0DE04),
            (300, 200, 0, 48, 0xC0DE05), (90, 60, 0, 16, 0xC0DE06), (30, 5, 0, 4, 0xC0DE07), (450, 375, 0, 64, 0xC0DE08),
            (200, 120, 0, 32, 0xC0DE09)]


def main():
    from oracle.pyoracle import Oracle, Reference, default_option, sha
    orc = Oracle()
    ref = Reference.for_shape(450, 375, 64)
    assert ref is not None, "oracle/build_ref.sh 450 375 64 first"
    pairs = [orc.synth_pair(w, h, dmax - dmin, seed) for (w, h, dmin, dmax, seed) in SEQUENCE]
    ref.clear_census()                                        # a new process
    steps = []
    for (w, h, dmin, dmax, seed), (l, r) in zip(SEQUENCE, pairs):
        out = ref.api_match(l, r, default_option(dmax, dmin), reset=True, clear=False)
        assert out is not None
        steps.append({"w": w, "h": h, "dmin": dmin, "dmax": dmax, "seed": seed, "sha256_in_sequence": sha(out),
                      "sha256_inputs": {"left": sha(l), "right": sha(r)}})
    for st, (w, h, dmin, dmax, seed), (l, r) in zip(steps, SEQUENCE, pairs):
        out = ref.api_match(l, r, default_option(dmax, dmin), reset=True, clear=True)
        st["sha256_fresh"] = sha(out)
        st["history_matters"] = st["sha256_fresh"] != st["sha256_in_sequence"]
    assert any(s["history_matters"] for s in steps)
    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/make_golden_history.py", "reference": os.path.basename(ref.path),
                   "options": "main.c:48-65 with [dmin, dmax) per step", "steps": steps}, f, indent=1)
    print("wrote", OUT, [s["history_matters"] for s in steps])


if __name__ == "__main__":
    main()
