#!/usr/bin/env python3
"""Golden digests for the 4-path mode (BASELINE config 0: "Cones 450x375, D=64, 4 paths"), made by the REFERENCE'S OWN stage functions.

The reference never reads `num_paths`, so it cannot be asked for four paths; the mode is defined as the FIRST FOUR of
its eight CostAggregate calls (SemiGlobalMatching.c:213-216: (1,0), (-1,0), (0,1), (0,-1)).  oracle/ref_harness_tail.c replays
SGM_Match with exactly those four calls -- every stage still the reference's own code -- and this script (build container only) hashes
what comes out.  Until round 4 the mode was pinned by our CPU restatement alone.

    python tests/golden/make_golden_paths4.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.pyoracle import STAGE_NAMES, Oracle, Reference, default_option, sha  # noqa: E402
from make_golden import opt_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    gen = Oracle()
    cases = []

    def add(name, left, right, opt, seed=None, inputs_file=None, note=""):
        h, w = left.shape
        d = opt.max_disparity - opt.min_disparity
        ref = Reference.for_shape(w, h, d)
        assert ref is not None, (w, h, d)
        st = ref.run(left, right, opt, first_dirs=4)
        e = {"name": name, "w": w, "h": h, "d": d, "seed": seed, "option": opt_dict(opt), "note": note, "first_dirs": 4,
             "oob_dropped": ref.oob_count(), "sha256": {n: sha(st[n]) for n in STAGE_NAMES},
             "sha256_inputs": {"left": sha(left), "right": sha(right)}, "invalid_final": int(np.isinf(st["final"]).sum())}
        if inputs_file:
            e["inputs_file"] = inputs_file
        cases.append(e)
        print(f"{name:26s} {w}x{h}x{d} invalid={e['invalid_final']}")

    with np.load(os.path.join(OUT, "cone_inputs.npz")) as z:
        add("p4_cone", z["left"], z["right"], default_option(64, num_paths=4), inputs_file="cone_inputs.npz",
            note="BASELINE config 0: Data/cone pair, main.c:48-65 options, four paths")
    with np.load(os.path.join(OUT, "scene_reindeer.npz")) as z:
        add("p4_scene_reindeer", z["left"], z["right"], default_option(128, num_paths=4), inputs_file="scene_reindeer.npz")
    for name, w, h, dmin, dmax, seed, kw in (("p4_t70x33_d16", 70, 33, 0, 16, 0x5EED5001, {"min_speckle_area": 12}),
                                             ("p4_t20x31_d8_tall", 20, 31, 0, 8, 0x5EED5002, {"min_speckle_area": 6}),
                                             ("p4_t40x24_d16_dmin3", 40, 24, 3, 19, 0x5EED5003, {"min_speckle_area": 8}),
                                             ("p4_kitti_1242x375_d128", 1242, 375, 0, 128, 0x5EED0002, {})):
        l, r = gen.synth_pair(w, h, dmax - dmin, seed)
        add(name, l, r, default_option(dmax, dmin, num_paths=4, **kw), seed=seed)
    with open(os.path.join(OUT, "cases_paths4.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_paths4.py",
                   "source": "oracle/_ref: the reference's stage functions, first four CostAggregate calls (ref_run_stages_first_dirs)",
                   "cases": cases}, f, indent=1)


if __name__ == "__main__":
    main()
