#!/usr/bin/env python3
"""How long does the host take to ENQUEUE frames (no sync) vs the GPU to finish them?  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import soc_project_stereo_matching_amd as S
w, h, d = 1242, 375, 128
opt = S.default_option(d)
for n_inst, timing in ((4, True), (4, False), (8, False)):
    insts = [S.SGMInstance(0) for _ in range(n_inst)]
    for i in insts:
        assert i.reset(w, h, opt)
        i.enable_timing(timing)
    l, r = S.synth_pair(w, h, d, 1)
    dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    outs = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(n_inst)]
    torch.cuda.synchronize()
    def step(k):
        i = insts[k % n_inst]
        i.reset(w, h, opt)
        i.match_device(dl.data_ptr(), dr.data_ptr(), outs[k % n_inst].data_ptr())
    for k in range(20): step(k)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for k in range(K): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"instances {n_inst} timing_events {timing}: host enqueue {1e3*(t1-t0)/K:.3f} ms/frame, total {1e3*(t2-t0)/K:.3f} ms/frame")
    for i in insts: i.close()
