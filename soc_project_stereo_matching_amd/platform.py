"""Test-platform arithmetic next to the hot path: disparity -> depth and the scores the
reference's server computes for a returned depth image.

Restated from reading HostScript_Server/depth_image.py (its module imports cv2, which is not installed here, so it
cannot be imported to generate vectors: "parity unpinned" for these two functions) and from client.py:40-45.
"""
from __future__ import annotations

import numpy as np


def disparity_to_depth(disp: np.ndarray, fx: float, baseline: float, doffs: float) -> np.ndarray:
    """depth[mm] = fx * baseline / (disparity + doffs) (depth_image.py:138-165); invalid (+inf / nan) disparities
    and a zero denominator give NaN, as in the platform's client simulator (client.py:40-45).  float32."""
    disp = np.asarray(disp, np.float32)
    # the board knows the calibration as the float32 values of the 80-byte block (stereo_calibration.py:177-195, client.py:31-37);
    # their product is formed in double and rounded once, like client.py:44's python-float product against a float32 array
    fx, baseline, doffs = np.float32(fx), np.float32(baseline), np.float32(doffs)
    denom = disp + doffs
    depth = np.full(disp.shape, np.nan, np.float32)
    ok = np.isfinite(denom) & (denom != 0)
    depth[ok] = np.float32(float(fx) * float(baseline)) / denom[ok]
    return depth


def compare_depth(ground_truth: np.ndarray, test: np.ndarray, abs_thresh: float = 10.0):
    """(rmse, bad_pixel_rate, n_valid) over pixels finite in both images (depth_image.py:276-319);
    (nan, nan, 0) if there is none.  Bad pixel = |error| > abs_thresh millimetres."""
    valid = np.isfinite(test) & np.isfinite(ground_truth)
    n = int(np.count_nonzero(valid))
    if n == 0:
        return float("nan"), float("nan"), 0
    diff = test[valid] - ground_truth[valid]
    rmse = float(np.sqrt(np.mean(np.square(diff))))
    bpr = float(np.count_nonzero(np.abs(diff) > abs_thresh) / n)
    return rmse, bpr, n


def board_gray(b: np.ndarray, g: np.ndarray, r: np.ndarray, weight_r: int = 76) -> np.ndarray:
    """The firmware's grey conversion (ZedBoard/.../src/stereo_matching.c:18-25): (76 r + 150 g + 29 b) >> 8;
    weight_r = 77 gives stb_image's (stb_image.h:1746-1749), the one behind main.c's image load."""
    return ((weight_r * r.astype(np.uint32) + 150 * g.astype(np.uint32) + 29 * b.astype(np.uint32)) >> 8).astype(np.uint8)
