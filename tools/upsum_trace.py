#!/usr/bin/env python3
"""Diagnostics for the fused last sweep (csrc/sgm_upsum.hip): run ONE batch of KITTI frames alone with SGM_UPSUM_TRACE set and print how
its row groups followed one another -- the distance between the starts of consecutive groups of a frame (the chain's hop), a group's
own duration, how long it waited for its first hand-over, and where the groups ran.  Usage: python tools/upsum_trace.py [rows] [wgs]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/upsum_trace.bin"
os.environ["SGM_UPSUM"] = "1"
os.environ["SGM_UPSUM_TRACE"] = path
if len(sys.argv) > 1:
    os.environ["SGM_UPSUM_ROWS"] = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["SGM_UPSUM_WGS"] = sys.argv[2]
import torch  # noqa: E402
import soc_project_stereo_matching_amd as S  # noqa: E402

w, h, d, B = 1242, 375, 128, 8
opt = S.default_option(d)
i = S.SGMInstance(0, batch=B)
ps = [S.synth_pair(w, h, d, 0x5EED0002 + j) for j in range(B)]
L = torch.from_numpy(np.stack([p[0] for p in ps])).cuda()
R = torch.from_numpy(np.stack([p[1] for p in ps])).cuda()
out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
for rep in range(3):
    assert i.reset(w, h, opt) and i.match_device(L.data_ptr(), R.data_ptr(), out.data_ptr()) and i.synchronize()
with open(path, "rb") as f:
    nb, ng, rows, _ = struct.unpack("4i", f.read(16))
    t = np.frombuffer(f.read(), np.uint64).reshape(nb, ng, 12)
t0 = t[..., 0].astype(np.int64)
tick = 0.01                                                      # 100 MHz -> us
start = (t0 - t0.min()) * tick
seen = (t[..., 1].astype(np.int64) - t0) * tick
dur = (t[..., 2].astype(np.int64) - t[..., 1].astype(np.int64)) * tick
hop = np.diff(start, axis=1)
print(f"rows per workgroup {rows}, {ng} groups per frame, {nb} frames")
print(f"launch span              {((t[..., 2].astype(np.int64).max() - t0.min()) * tick):9.1f} us")
print(f"hop (start k+1 - start k) mean {hop.mean():7.2f}  median {np.median(hop):7.2f}  p90 {np.percentile(hop, 90):7.2f} us")
print(f"wait for first hand-over mean {seen.mean():7.2f}  median {np.median(seen):7.2f} us")
print(f"group duration (steps)   mean {dur.mean():7.2f}  median {np.median(dur):7.2f}  min {dur.min():7.2f} us")
xcc = (t[..., 3] >> np.uint64(32)).astype(int)
print("frames x XCCs used:", [sorted(set(xcc[f].tolist())) for f in range(nb)])
fr = 0
print("frame 0, groups 0..11: start / first hand-over seen after / duration (us)")
for k in range(min(12, ng)):
    print(f"  {k:3d} {start[fr, k]:9.1f} {seen[fr, k]:8.2f} {dur[fr, k]:8.2f}  xcc {xcc[fr, k]}")
steps = (w - 1 + d - 1 + rows - 1) // 16 + 1 + rows - 1
for name, o in (("bottom team", 4), ("top team", 8)):
    ph = t[..., o:o + 4].astype(np.float64) / steps                  # shader clocks per step
    print(f"{name}, wave 0, shader clocks per step (mean over groups): work before barrier A {ph[..., 0].mean():7.0f}, wait at A {ph[..., 1].mean():7.0f}, "
          f"work before B {ph[..., 2].mean():7.0f}, wait at B {ph[..., 3].mean():7.0f}   | group 0 of frame 0: {ph[0, 0].round().tolist()}")
i.close()
