#!/usr/bin/env python3
"""Shader clock during the aggregation launches of a running benchmark loop (diagnostics, on the GPU box):
batches of 8 KITTI frames, two instances in flight for a few seconds, then s_memtime / s_memrealtime of the last launch.
Needs a build with the probe compiled in (the product build has none):
    tools/build_variant.sh clock CLOCK_PROBE=1 && SGM_LIBRARY_PATH=variants/clock/libsgm_mi355x.so python tools/agg_clock.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import soc_project_stereo_matching_amd as S  # noqa: E402

w, h, d, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1242, 375, 128, 8)))
opt = S.default_option(d)
for n_inst in ((1,) if len(sys.argv) > 4 else (1, 2)):
    insts = [S.SGMInstance(0, batch=B) for _ in range(n_inst)]
    ps = [S.synth_pair(w, h, d, 0x5EED0002 + j) for j in range(B)]
    l = torch.from_numpy(np.stack([p[0] for p in ps])).cuda()
    r = torch.from_numpy(np.stack([p[1] for p in ps])).cuda()
    outs = [torch.empty((B, h, w), dtype=torch.float32, device="cuda") for _ in insts]
    for i in insts:
        assert i.reset(w, h, opt)
    t0 = time.time()
    k = 0
    while time.time() - t0 < 4.0:
        i = insts[k % n_inst]
        i.reset(w, h, opt)
        i.match_device(l.data_ptr(), r.data_ptr(), outs[k % n_inst].data_ptr())
        k += 1
        if k % 64 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 2)()
    if S.load_library().sgmd_debug_clock(0, out) != 0:
        raise SystemExit("this library was built without the clock probe: tools/build_variant.sh clock CLOCK_PROBE=1")
    i0 = insts[0]
    i0.enable_timing(True)
    for _ in range(3):
        i0.reset(w, h, opt); i0.match_device(l.data_ptr(), r.data_ptr(), outs[0].data_ptr())
    i0.synchronize()
    print("aggregation launch (events, alone):", round(i0.mean_timing()[0]["aggregate"], 3), "ms")
    S.load_library().sgmd_debug_clock(0, out)
    print(f"{n_inst} batch(es) in flight, {k} steps in {time.time() - t0:.1f} s: block 0 of the last aggregation launch lived "
          f"{out[1] / 100.0:.1f} us = {out[0]} shader cycles -> {out[0] / out[1] * 0.1:.3f} GHz", flush=True)
    for i in insts:
        i.close()
