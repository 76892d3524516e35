#!/usr/bin/env python3
"""The workload the rocprofv3 counter passes of tools/profile_counters.py wrap: a few sgm_reset + sgm_match_device
passes of one instance over HBM-resident synthetic frames (the bench's step, without timing or verification).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o x --output-format csv -- python3 tools/pmc_workload.py --batch 1
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="kitti_1242x375_d128_p8", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import torch
    import soc_project_stereo_matching_amd as S
    w, h, d, seed = WORKLOADS[args.workload]
    opt = S.default_option(d)
    B = args.batch
    inst = S.SGMInstance(0, batch=B)
    assert inst.reset(w, h, opt)
    ps = [S.synth_pair(w, h, d, seed + j) for j in range(B)]
    l = torch.from_numpy(np.stack([p[0] for p in ps])).cuda()
    r = torch.from_numpy(np.stack([p[1] for p in ps])).cuda()
    out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(args.steps):
        assert inst.reset(w, h, opt)
        assert inst.match_device(l.data_ptr(), r.data_ptr(), out.data_ptr())
        inst.synchronize()
    inst.close()


if __name__ == "__main__":
    main()
