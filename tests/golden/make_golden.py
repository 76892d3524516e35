#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Every expected value written here is produced by oracle/_ref/libsgm_ref_*.so, i.e. the
reference's own SemiGlobalMatching.c compiled by oracle/build_ref.sh (guarded build, SURVEY.md
8c).  Our CPU restatement is NOT used to produce expectations (only its seeded input generator
is used for synthetic pairs, and the inputs themselves are stored or re-derivable from the seed).

Outputs
  cases.json            one entry per case: shape, options, seed, sha256 of every stage
  tiny_<name>.npz       full input + all nine stage arrays for the tiny cases
  cone_inputs.npz       the cone pair as 8-bit grey (stb formula) + the committed im2.d.png
  cone_final.npz        final float disparity of the cone pair
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import (DIRECTIONS, STAGE_NAMES, Oracle, Reference, default_option,  # noqa: E402
                             load_gray_stb, sha)

OUT = os.path.dirname(os.path.abspath(__file__))
CONE = "/root/reference/SemiGlobalMatching/Data/cone/"

OPT_FIELDS = ["num_paths", "min_disparity", "max_disparity", "is_check_unique", "uniqueness_ratio",
              "is_check_lr", "lrcheck_thres", "is_remove_speckles", "min_speckle_area", "p1", "p2_init"]


def opt_dict(o):
    return {k: (float(getattr(o, k)) if k in ("uniqueness_ratio", "lrcheck_thres") else int(getattr(o, k)))
            for k in OPT_FIELDS}


def main():
    gen = Oracle()           # input generator only
    cases = []

    def run_case(name, left, right, opt, seed=None, store=False, note=""):
        h, w = left.shape
        d = opt.max_disparity - opt.min_disparity
        ref = Reference.for_shape(w, h, d)
        assert ref is not None, f"no reference build covers {w}x{h}x{d}; run `make -C oracle ref`"
        st = ref.run(left, right, opt)
        entry = {"name": name, "w": w, "h": h, "d": d, "seed": seed, "option": opt_dict(opt), "note": note,
                 "oob_dropped": ref.oob_count(),
                 "sha256": {n: sha(st[n]) for n in STAGE_NAMES},
                 "invalid_final": int(np.isinf(st["final"]).sum())}
        if seed is not None:
            entry["sha256_inputs"] = {"left": sha(left), "right": sha(right)}
        if store:
            np.savez_compressed(os.path.join(OUT, f"tiny_{name}.npz"), left=left, right=right, **st)
            entry["file"] = f"tiny_{name}.npz"
        cases.append(entry)
        print(f"{name:28s} {w}x{h}x{d} invalid={entry['invalid_final']} oob={entry['oob_dropped']}")
        return st

    # ---- cone pair: the reference's only self-reproduced fixture ----
    from PIL import Image
    left = load_gray_stb(CONE + "im2.png")
    right = load_gray_stb(CONE + "im6.png")
    committed_png = np.asarray(Image.open(CONE + "im2.d.png"), np.uint8)
    np.savez_compressed(os.path.join(OUT, "cone_inputs.npz"), left=left, right=right, im2_d_png=committed_png)
    st = run_case("cone", left, right, default_option(64), note="main.c:48-65 options on Data/cone/im2.png, im6.png")
    np.savez_compressed(os.path.join(OUT, "cone_final.npz"), final=st["final"])

    # ---- tiny cases with full arrays ----
    def synth(w, h, d, seed):
        return gen.synth_pair(w, h, d, seed)

    tiny = [
        ("t24x16_d8", 24, 16, 0, 8, {"min_speckle_area": 6}),
        ("t70x33_d16", 70, 33, 0, 16, {"min_speckle_area": 12}),
        ("t20x31_d8_tall", 20, 31, 0, 8, {"min_speckle_area": 6}),          # W < H: early wraps (Q5)
        ("t40x24_d16_dmin3", 40, 24, 3, 19, {"min_speckle_area": 8}),       # min_disparity > 0
        ("t33x33_d12_square", 33, 33, 0, 12, {"min_speckle_area": 6}),      # W == H, D not a multiple of 4
        ("t64x20_d40", 64, 20, 0, 40, {"min_speckle_area": 10, "p1": 7, "p2_init": 99}),
    ]
    for i, (name, w, h, dmin, dmax, kw) in enumerate(tiny):
        seed = 0x5EED1000 + i
        l, r = synth(w, h, dmax - dmin, seed)
        run_case(name, l, r, default_option(dmax, dmin, **kw), seed=seed, store=True)

    # ---- option variants on one small shape (digests only; inputs from the seed) ----
    seed = 0x5EED2000
    l, r = synth(96, 40, 32, seed)
    variants = {
        "v_default": {},
        "v_no_unique": {"is_check_unique": False},
        "v_no_lr": {"is_check_lr": False},
        "v_no_speckle": {"is_remove_speckles": False},
        "v_plain": {"is_check_unique": False, "is_check_lr": False, "is_remove_speckles": False},
        "v_p1_0_p2_0": {"p1": 0, "p2_init": 0},
        "v_p2_small": {"p1": 20, "p2_init": 8},
        "v_p_big": {"p1": 60, "p2_init": 250},
        "v_ratio_095": {"uniqueness_ratio": 0.95},
        "v_lr_thres_0": {"lrcheck_thres": 0.0},
        "v_speckle_area_400": {"min_speckle_area": 400},
        "v_num_paths_4_ignored": {"num_paths": 4},        # Q1: the reference ignores num_paths
    }
    for name, kw in variants.items():
        run_case(name, l, r, default_option(32, 0, **kw), seed=seed)

    # ---- config shapes of BASELINE.json (digests only) ----
    seed = 0x5EED0001
    l, r = synth(450, 375, 64, seed)
    run_case("c1_synth_450x375_d64", l, r, default_option(64), seed=seed, note="BASELINE config 0 shape, 8 paths")
    seed = 0x5EED0002
    l, r = synth(1242, 375, 128, seed)
    run_case("c2_kitti_1242x375_d128", l, r, default_option(128), seed=seed, note="BASELINE config 1 (headline)")
    seed = 0x5EED0003
    l, r = synth(400, 48, 256, seed)
    run_case("d256_400x48", l, r, default_option(256, min_speckle_area=20), seed=seed,
             note="D=256 needs the widened loop counter (Q2)")
    seed = 0x5EED0004
    l, r = synth(300, 60, 192, seed)
    run_case("d192_300x60", l, r, default_option(192, min_speckle_area=20), seed=seed, note="config 4 disparity range")

    # ---- per-direction aggregation on a tiny shape (pins Q4-Q6 per direction) ----
    seed = 0x5EED3000
    l, r = synth(37, 21, 8, seed)
    opt = default_option(8)
    ref = Reference.for_shape(37, 21, 8)
    st = ref.run(l, r, opt)
    per_dir = {}
    for dx, dy in DIRECTIONS:
        per_dir[f"S_{dx}_{dy}"] = ref.aggregate_dir(l, st["cost"], opt, dx, dy)
    np.savez_compressed(os.path.join(OUT, "perdir_37x21_d8.npz"), left=l, right=r, cost=st["cost"], **per_dir)
    l2, r2 = synth(15, 29, 8, seed + 1)          # W < H
    ref2 = Reference.for_shape(15, 29, 8)
    st2 = ref2.run(l2, r2, opt)
    per_dir2 = {f"S_{dx}_{dy}": ref2.aggregate_dir(l2, st2["cost"], opt, dx, dy) for dx, dy in DIRECTIONS}
    np.savez_compressed(os.path.join(OUT, "perdir_15x29_d8.npz"), left=l2, right=r2, cost=st2["cost"], **per_dir2)

    # ---- Q14: second Match without Reset accumulates onto the previous S ----
    seed = 0x5EED4000
    l, r = synth(48, 20, 16, seed)
    l2, r2 = synth(48, 20, 16, seed + 1)
    opt = default_option(16, min_speckle_area=8)
    ref = Reference.for_shape(48, 20, 16)
    first = ref.api_match(l, r, opt, reset=True)
    second = ref.api_match(l2, r2, opt, reset=False)       # S still holds frame 1's sums
    second_fresh = ref.api_match(l2, r2, opt, reset=True)
    np.savez_compressed(os.path.join(OUT, "q14_no_reset_48x20_d16.npz"), left=l, right=r, left2=l2, right2=r2,
                        first=first, second=second, second_fresh=second_fresh)

    # ---- test-platform calibration block: produced by the reference's own
    # HostScript_Server/stereo_calibration.py (numpy only; imported here, never shipped) ----
    sys.path.insert(0, "/root/reference/HostScript_Server")
    import stereo_calibration                      # noqa: E402  (reference module, build container only)
    calib_txt = ("cam0=[1733.74 0 792.27; 0 1733.74 541.89; 0 0 1]\n"
                 "cam1=[1733.74 0 792.27; 0 1733.74 541.89; 0 0 1]\n"
                 "doffs=0\nbaseline=536.62\nwidth=1920\nheight=1080\nndisp=170\n")
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as tf:
        tf.write(calib_txt)
    cal = stereo_calibration.StereoCalib(tf.name)
    cal.scale_calib(1280, 720)
    packed = np.frombuffer(cal.pack(), np.uint8).copy()
    os.unlink(tf.name)
    np.savez_compressed(os.path.join(OUT, "platform_calib.npz"), calib_txt=np.frombuffer(calib_txt.encode(), np.uint8),
                        packed=packed, fx=np.float64(cal.cam0[0, 0]), doffs=np.float64(cal.doffs),
                        baseline=np.float64(cal.baseline))

    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": "oracle/_ref (reference C, guarded build)",
                   "cases": cases}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
