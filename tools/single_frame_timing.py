#!/usr/bin/env python3
"""Per-stage device time of ONE frame per launch (the latency configuration) for a few lane layouts.
GPU box:  python tools/single_frame_timing.py [workload]      (env knobs are read at sgm_reset)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import soc_project_stereo_matching_amd as S   # noqa: E402

w, h, d = (1242, 375, 128) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
left, right = S.synth_pair(w, h, d, 0x5EED0002)
opt = S.default_option(d)
want = None
ENVS = ({},) if os.environ.get("SGM_SINGLE_ONLY_DEFAULT") else None
for env in ENVS or ({}, {"SGM_HL": "32"}, {"SGM_HL": "0"}, {"SGM_HL": "64", "SGM_LANES_PER_PIXEL": "8"}, {"SGM_HL": "32", "SGM_LANES_PER_PIXEL": "8"},
            {"SGM_FUSED_WTA": "1"}):
    for k in ("SGM_HL", "SGM_LANES_PER_PIXEL", "SGM_FUSED_WTA"):
        os.environ.pop(k, None)
    os.environ.update(env)
    i = S.SGMInstance(0)
    i.enable_timing(True)
    outs = []
    for _ in range(8):
        assert i.reset(w, h, opt)
        outs.append(i.match(left, right))
    i.enable_timing(True)                    # new statistics window after the warm-up frames
    for _ in range(10):
        assert i.reset(w, h, opt)
        out = i.match(left, right)
    mean, mn, cnt = i.mean_timing()
    if want is None:
        want = out
    same = np.array_equal(out.view(np.uint32), want.view(np.uint32))
    tot = sum(mean.values())
    print(env or "default", "same" if same else "DIFFERENT", f"total {tot:.3f} ms",
          {k: round(v, 4) for k, v in mean.items()}, flush=True)
    i.close()
