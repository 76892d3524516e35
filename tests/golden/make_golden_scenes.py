#!/usr/bin/env python3
"""Golden vectors for the three image pairs the reference ships beside `cone` and never runs itself
(SemiGlobalMatching/Data/{Cloth3,Reindeer,Wood2}: view1.png / view5.png, 626-671 x 555 RGB, drange.txt 0..128;
main.c:19-20 hard-codes cone) -- real texture / occlusion statistics (long speckle components, uniqueness failures,
invalid bands) instead of the LCG-synthetic pairs of every other case beyond cone.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden_scenes.py

Every expected value is produced by the REFERENCE's own C (oracle/build_ref.sh: guarded build at capacity 700x560x128, as for
every other golden) with the options of main.c:48-65 and max_disparity = 128 (drange.txt); the grey images come from the stb
formula main.c's loader applies (oracle.pyoracle.load_gray_stb).  Committed: the grey inputs (scene_<name>.npz), every stage's
SHA-256 (cases_scenes.json), the final map of one scene, and one scene's RGB arrays (the colour -> grey reader of csrc/sgm_main.c
is then exercised at a second size).  Images are data files of the reference, stored as arrays -- no source text.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import STAGE_NAMES, Reference, default_option, load_gray_stb, sha  # noqa: E402
from make_golden import opt_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DATA = "/root/reference/SemiGlobalMatching/Data/"
SCENES = ["Cloth3", "Reindeer", "Wood2"]
RGB_SCENE = "Reindeer"                      # widest of the three; its RGB arrays feed the sgm_main test


def main():
    from PIL import Image
    subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh"), "700", "560", "128"])
    cases = []
    for scene in SCENES:
        with open(DATA + scene + "/drange.txt") as f:
            rng = dict(ln.strip().split("=") for ln in f if "=" in ln)
        dmin, dmax = int(rng["dmin"]), int(rng["dmax"])
        left = load_gray_stb(DATA + scene + "/view1.png")
        right = load_gray_stb(DATA + scene + "/view5.png")
        h, w = left.shape
        opt = default_option(dmax, dmin)
        ref = Reference.for_shape(w, h, dmax - dmin)
        assert ref is not None and ref.capacity[0] >= w
        st = ref.run(left, right, opt)
        name = "scene_" + scene.lower()
        extra = {}
        if scene == RGB_SCENE:
            extra = {"rgb_left": np.asarray(Image.open(DATA + scene + "/view1.png").convert("RGB"), np.uint8),
                     "rgb_right": np.asarray(Image.open(DATA + scene + "/view5.png").convert("RGB"), np.uint8),
                     "final": st["final"]}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), left=left, right=right, **extra)
        entry = {"name": name, "w": w, "h": h, "d": dmax - dmin, "seed": None, "option": opt_dict(opt),
                 "note": f"Data/{scene}/view1.png, view5.png (stb grey), main.c:48-65 options, drange.txt {dmin}..{dmax}",
                 "oob_dropped": ref.oob_count(), "inputs_file": name + ".npz",
                 "sha256": {n: sha(st[n]) for n in STAGE_NAMES},
                 "sha256_inputs": {"left": sha(left), "right": sha(right)},
                 "invalid_final": int(np.isinf(st["final"]).sum()),
                 "invalid_after_lr": int(np.isinf(st["after_lr"]).sum()) if "after_lr" in st else None}
        cases.append(entry)
        print(f"{name:16s} {w}x{h}x{dmax - dmin} invalid={entry['invalid_final']} oob={entry['oob_dropped']}")
    with open(os.path.join(OUT, "cases_scenes.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_scenes.py", "source": "oracle/_ref (reference C, guarded build, capacity 700x560x128)",
                   "cases": cases}, f, indent=1)


if __name__ == "__main__":
    main()
