#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8(f)-3 (disparity -> depth, RMSE / bad-pixel scoring) produced by the REFERENCE'S OWN FUNCTIONS.

HostScript_Server/depth_image.py cannot be imported here: its first lines import cv2, which this image does not have (an ordinary
ModuleNotFoundError; no stand-in for cv2 is made).  But the two functions on the path -- `disparity_to_depth` (:138-165) and
`compare_img` (:276-319) -- are plain numpy.  This script (build container only) parses the file with `ast`, compiles exactly those two
FunctionDef nodes from the reference's text where it lies, and runs them; the calibration object is the reference's own
stereo_calibration.StereoCalib (numpy only, imported as make_golden.py already does).  Nothing of the reference is copied: the repo
gets arrays -- inputs and what the reference's functions returned for them (tests/golden/platform_depth.npz).

    python tests/golden/make_golden_depth.py
"""
import ast
import logging
import os
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/HostScript_Server"


def reference_functions():
    sys.path.insert(0, REF)
    import stereo_calibration                                   # noqa: E402  reference module (numpy only), never shipped
    with open(os.path.join(REF, "depth_image.py")) as f:
        tree = ast.parse(f.read())
    ns = {"np": np, "logging": logging, "StereoCalib": stereo_calibration.StereoCalib}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("disparity_to_depth", "compare_img"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "depth_image.py", "exec"), ns)
    return stereo_calibration.StereoCalib, ns["disparity_to_depth"], ns["compare_img"]


def main():
    StereoCalib, ref_depth, ref_compare = reference_functions()
    log = logging.getLogger("golden")
    out = {"numpy_version": np.frombuffer(np.__version__.encode(), np.uint8)}
    calibs = {
        # the block the platform tests use (doffs 0) and a Middlebury-style one with a non-zero doffs
        "a": "cam0=[1733.74 0 792.27; 0 1733.74 541.89; 0 0 1]\ncam1=[1733.74 0 792.27; 0 1733.74 541.89; 0 0 1]\ndoffs=0\nbaseline=536.62\nwidth=1920\nheight=1080\nndisp=170\n",
        "b": "cam0=[3979.911 0 1244.772; 0 3979.911 1019.507; 0 0 1]\ncam1=[3979.911 0 1369.115; 0 3979.911 1019.507; 0 0 1]\ndoffs=124.343\nbaseline=193.001\nwidth=2964\nheight=1988\nndisp=280\n",
    }
    rng = np.random.default_rng(20261005)
    with np.load(os.path.join(OUT, "cone_final.npz")) as z:
        cone = z["final"]
    with np.load(os.path.join(OUT, "scene_reindeer.npz")) as z:
        reindeer = z["final"]
    maps = {
        "cone": cone,                                            # the reference's own output incl. +INF (invalid)
        "reindeer": reindeer[100:300, 200:520].copy(),
        "random": (rng.random((64, 97), dtype=np.float32) * 200).astype(np.float32),
        "edge": np.array([[0.0, -0.0, 1.0, 0.5, 127.75, 3.4028235e38, 1e-30, -5.0, np.inf, -124.343, 63.99999]], np.float32),
    }
    for cname, txt in calibs.items():
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as tf:
            tf.write(txt)
        cal = StereoCalib(tf.name)
        cal.scale_calib(1280, 720)
        os.unlink(tf.name)
        out[f"calib_{cname}"] = np.array([cal.cam0[0, 0], cal.baseline, cal.doffs], np.float64)
        out[f"calib_{cname}_fx_is_float32"] = np.array([isinstance(cal.cam0[0, 0], np.float32)])
        for mname, m in maps.items():
            if cname == "b" and mname == "cone":
                continue                                         # (keeps the fixture small)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                d = ref_depth(m, cal)
            assert d.dtype == np.float32, d.dtype
            out[f"disp_{mname}"] = m
            out[f"depth_{cname}_{mname}"] = d
    # scoring: ground truth = the reference's depth of a map, test = the same map perturbed, with non-finite pixels on either side
    k = 0
    for mname in ("cone", "reindeer", "random"):
        gt = out[f"depth_a_{mname}"][:160, :320].copy()
        test = gt + rng.normal(0, 6, gt.shape).astype(np.float32)
        test[rng.random(gt.shape) < 0.05] = np.nan
        test[rng.random(gt.shape) < 0.02] = np.inf
        gt[rng.random(gt.shape) < 0.03] = np.nan
        out[f"score_{k}_gt"], out[f"score_{k}_test"] = gt, test
        res = []
        for thr in (10, 2.5):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                rmse, bpr, n = ref_compare(gt, test, log, thr)
            res.append([rmse, bpr, n, thr])
        out[f"score_{k}_results"] = np.array(res, np.float64)
        k += 1
    empty = np.full((4, 5), np.nan, np.float32)
    r = ref_compare(empty, empty, log)
    out["score_empty_result"] = np.array([r[0], r[1], r[2]], np.float64)
    out["n_scores"] = np.array([k])
    np.savez_compressed(os.path.join(OUT, "platform_depth.npz"), **out)
    print("wrote platform_depth.npz:", k, "score cases,", len(maps) * len(calibs), "depth maps; numpy", np.__version__)


if __name__ == "__main__":
    main()
