"""The CPU checker of the test-platform arithmetic (oracle/platform_oracle.py) against values worked out by hand from the
reference's published formulas (depth_image.py:138-165, :276-319; stereo_matching.c:18-25).  No reference-made vector exists
for these (cv2 is not installed, nothing recorded in the reference): "parity unpinned" -- this pins the checker to the
formulas, the GPU tests pin the device kernels to the checker."""
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


def test_board_gray_known_answers():
    b = np.array([0, 255, 10, 200], np.uint8)
    g = np.array([0, 255, 20, 100], np.uint8)
    r = np.array([0, 255, 30, 50], np.uint8)
    assert board_gray(b, g, r, 76).tolist() == [0, (255 * 255) >> 8, (76 * 30 + 150 * 20 + 29 * 10) >> 8, (76 * 50 + 150 * 100 + 29 * 200) >> 8]
    assert board_gray(b, g, r, 77).tolist() == [0, 255, (77 * 30 + 150 * 20 + 29 * 10) >> 8, (77 * 50 + 150 * 100 + 29 * 200) >> 8]
