"""The CPU checker of the test-platform arithmetic (oracle/platform_oracle.py) against values worked out by hand from the
reference's published formulas (depth_image.py:138-165, :276-319; stereo_matching.c:18-25) and -- round 4 -- against vectors the
REFERENCE'S OWN FUNCTIONS produced: depth_image.py cannot be imported (cv2), but `disparity_to_depth` and `compare_img` are plain
numpy, and tests/golden/make_golden_depth.py compiles just those two definitions from the reference's text and runs them
(tests/golden/platform_depth.npz).  The grey conversion stays pinned to the formula only (the firmware file needs Xilinx headers)."""
import math

import numpy as np

from oracle.platform_oracle import board_gray, compare_depth, disparity_to_depth


def test_depth_known_answers():
    disp = np.array([[10.0, 0.0, np.inf], [-2.5, 30.5, 100.0]], np.float32)
    got = disparity_to_depth(disp, fx=1000.0, baseline=200.0, doffs=2.5)
    assert got.dtype == np.float32
    # 200000 / (d + 2.5): 16000, 80000, nan (invalid disparity), nan (zero denominator), 6060.606..., 1951.2195...
    want = [np.float32(200000.0) / np.float32(12.5), np.float32(200000.0) / np.float32(2.5), np.nan, np.nan,
            np.float32(200000.0) / np.float32(33.0), np.float32(200000.0) / np.float32(102.5)]
    for g, w in zip(got.ravel(), want):
        assert (math.isnan(g) and math.isnan(w)) or g == w
    assert got[0, 0] == 16000.0 and got[0, 1] == 80000.0


def test_scores_known_answers():
    gt = np.array([1000.0, 2000.0, np.nan, 4000.0, 5000.0, np.inf], np.float32)
    te = np.array([1003.0, 1980.0, 3000.0, np.nan, 5011.0, 1.0], np.float32)
    rmse, bpr, n = compare_depth(gt, te, abs_thresh=10.0)
    # valid pairs: errors +3, -20, +11 -> rmse = sqrt((9 + 400 + 121) / 3), two of three beyond 10 mm
    assert n == 3
    assert abs(rmse - math.sqrt(530.0 / 3.0)) < 1e-12
    assert bpr == 2.0 / 3.0
    r, b, n0 = compare_depth(np.full(4, np.nan, np.float32), np.ones(4, np.float32))
    assert math.isnan(r) and math.isnan(b) and n0 == 0
    assert compare_depth(gt, gt)[:2] == (0.0, 0.0)
    # exactly at the threshold is not bad (strict >)
    assert compare_depth(np.array([0.0], np.float32), np.array([10.0], np.float32), 10.0) == (10.0, 0.0, 1)


def _depth_cases():
    from conftest import load_npz
    z = load_npz("platform_depth.npz")
    for c in ("a", "b"):
        fx, baseline, doffs = (float(v) for v in z[f"calib_{c}"])
        assert bool(z[f"calib_{c}_fx_is_float32"][0])           # the reference keeps the camera matrix as float32: float32 arithmetic
        for m in ("cone", "reindeer", "random", "edge"):
            if f"depth_{c}_{m}" in z:
                yield f"{c}/{m}", z[f"disp_{m}"], fx, baseline, doffs, z[f"depth_{c}_{m}"]


def check_depth_against_reference(got, disp, doffs, ref, what):
    """Bit-identical wherever the denominator disparity + doffs is finite and not zero.  Elsewhere the library answers NaN by design
    (an invalid disparity is +INF here; client.py:40-45 masks invalid pixels to NaN) where the reference's bare formula gives 0 (for
    +INF) or +-inf (zero denominator): asserted as exactly that, so the two can only differ in pixels compare_img ignores or that
    carry the invalid marker."""
    denom = disp + np.float32(doffs)
    usable = np.isfinite(denom) & (denom != 0)
    assert got.dtype == np.float32 and ref.dtype == np.float32
    assert np.array_equal(got[usable].view(np.uint32), ref[usable].view(np.uint32)), what
    assert np.isnan(got[~usable]).all(), what
    zero_den = np.isfinite(denom) & (denom == 0)
    assert np.isinf(ref[zero_den]).all() and (ref[np.isinf(denom)] == 0).all(), what


def test_depth_equals_the_references_own_function():
    n = 0
    for what, disp, fx, baseline, doffs, ref in _depth_cases():
        check_depth_against_reference(disparity_to_depth(disp, fx, baseline, doffs), disp, doffs, ref, what)
        n += 1
    assert n == 7


def test_scores_equal_the_references_own_function():
    from conftest import load_npz
    z = load_npz("platform_depth.npz")
    for k in range(int(z["n_scores"][0])):
        gt, te = z[f"score_{k}_gt"], z[f"score_{k}_test"]
        for rmse_r, bpr_r, n_r, thr in z[f"score_{k}_results"]:
            rmse, bpr, n = compare_depth(gt, te, float(thr))
            assert n == int(n_r) and bpr == bpr_r, (k, thr)
            # the reference's np.mean over float32 squares accumulates pairwise in float32; this checker (and the device) in float64
            assert abs(rmse - rmse_r) <= 2e-6 * rmse_r, (k, thr, rmse, rmse_r)
    r = compare_depth(np.full((4, 5), np.nan, np.float32), np.full((4, 5), np.nan, np.float32))
    assert math.isnan(r[0]) and math.isnan(r[1]) and r[2] == 0 and math.isnan(z["score_empty_result"][0]) and z["score_empty_result"][2] == 0


def test_board_gray_known_answers():
    b = np.array([0, 255, 10, 200], np.uint8)
    g = np.array([0, 255, 20, 100], np.uint8)
    r = np.array([0, 255, 30, 50], np.uint8)
    assert board_gray(b, g, r, 76).tolist() == [0, (255 * 255) >> 8, (76 * 30 + 150 * 20 + 29 * 10) >> 8, (76 * 50 + 150 * 100 + 29 * 200) >> 8]
    assert board_gray(b, g, r, 77).tolist() == [0, 255, (77 * 30 + 150 * 20 + 29 * 10) >> 8, (77 * 50 + 150 * 100 + 29 * 200) >> 8]
