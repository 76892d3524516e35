"""TEST INFRASTRUCTURE ONLY -- CPU checker for the test-platform arithmetic either side of the hot path.

Nothing in the product imports this module; the product's implementations are the device kernels sgm_gray_planes_k,
sgm_depth_k and sgm_score_k (csrc/sgm_post.hip) behind sgm_gray_from_planes / sgm_disparity_to_depth / sgm_compare_depth.
Only tests/ and the test tools under tools/ use it, as the thing the device results are compared with.

**Parity: depth and scoring pinned since round 4, the grey conversion unpinned.**  The reference holds these formulas in
HostScript_Server/depth_image.py (disparity_to_depth :138-165, compare_img :276-319), client.py:40-45 (the board simulator's depth
conversion) and, for the grey conversion, in the firmware (ZedBoard/Vitis/lwip_tcp_perf_client/src/stereo_matching.c:18-25).
depth_image.py and client.py import cv2, which is not installed here (an ordinary ModuleNotFoundError), and the firmware file needs
Xilinx headers.  But disparity_to_depth and compare_img themselves are plain numpy: tests/golden/make_golden_depth.py compiles just
those two definitions from the reference's text and runs them on the reference's own StereoCalib; tests/test_platform_oracle.py
checks this restatement against what they returned (tests/golden/platform_depth.npz): bit-identical depth wherever the denominator is
finite and non-zero, identical valid counts and bad-pixel rates, RMSE within 2e-6.  The grey formula has no reference-made vector.

Assumed arithmetic (stated because NumPy 1.x and 2.x promote `python float * float32 array` differently):
  * depth: the calibration reaches the board as float32 values (the 80-byte block, stereo_calibration.py:177-195); the
    product fx * baseline is formed once, rounded to float32, and divided by the float32 sum disparity + doffs -- one
    correctly rounded float32 divide per pixel, result float32.  A non-finite or zero denominator gives NaN
    (client.py:40-45 masks invalid disparities to NaN; depth_image.py:161 would produce 0 or inf there).
  * scores: the per-pixel error is the float32 difference of the two float32 images; squares and sums are taken in float64
    (the reference's np.mean over a float32 array accumulates pairwise in float32, so its RMSE can differ from this one in
    the last float32 digits; tests compare RMSE to 1e-6 relative).
"""
from __future__ import annotations

import math

import numpy as np


def board_gray(b, g, r, weight_r=76):
    """Integer grey of three colour planes: (weight_r * R + 150 * G + 29 * B) >> 8, weight_r = 76 for the firmware
    (stereo_matching.c:18-25) or 77 for stb_image's conversion behind main.c's image load (stb_image.h:1746-1749)."""
    acc = np.asarray(r, np.uint32) * np.uint32(weight_r)
    acc = acc + np.asarray(g, np.uint32) * np.uint32(150)
    acc = acc + np.asarray(b, np.uint32) * np.uint32(29)
    return np.right_shift(acc, 8).astype(np.uint8)


def disparity_to_depth(disp, fx, baseline, doffs):
    """Depth in millimetres of a float32 disparity map, float32 arithmetic as described in the module header."""
    d = np.asarray(disp, dtype=np.float32)
    scale = np.float32(float(np.float32(fx)) * float(np.float32(baseline)))       # one rounding of the product
    denom = d + np.float32(doffs)
    usable = np.isfinite(denom) & (denom != 0)
    out = np.full(d.shape, np.nan, dtype=np.float32)
    np.divide(scale, denom, out=out, where=usable)
    return out


def compare_depth(ground_truth, test, abs_thresh=10.0):
    """(rmse, bad_pixel_rate, n_valid) over the pixels that are finite in both depth images; (nan, nan, 0) when there is no
    such pixel.  A pixel is bad when its absolute error exceeds abs_thresh millimetres."""
    gt = np.asarray(ground_truth)
    te = np.asarray(test)
    both = np.isfinite(gt) & np.isfinite(te)
    n = int(both.sum())
    if n == 0:
        return math.nan, math.nan, 0
    err = (te[both] - gt[both]).astype(np.float64)            # float32 subtraction first (float32 inputs), then widened
    sum_sq = float(np.dot(err, err))
    n_bad = int((np.abs(err) > abs_thresh).sum())
    return math.sqrt(sum_sq / n), n_bad / n, n
