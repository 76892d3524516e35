#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path with the CPU oracle on a real MI355X.
Run on the GPU box:  python tools/gpu_diag.py [out.txt]
Diagnostic only (tests/ holds the actual parity tests)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import soc_project_stereo_matching_amd as S          # noqa: E402
from oracle.pyoracle import DIRECTIONS, STAGE_NAMES, Oracle, default_option   # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout


def log(*a):
    print(*a, file=out, flush=True)
    if out is not sys.stdout:
        print(*a, flush=True)


def feq(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else np.array_equal(a, b)


def run_case(inst, orc, w, h, dmin, dmax, seed, **kw):
    d = dmax - dmin
    l, r = orc.synth_pair(w, h, d, seed)
    opt = default_option(dmax, dmin, **kw)
    t0 = time.time()
    ref = orc.run(l, r, opt)
    t_cpu = time.time() - t0
    inst.keep_stages(True)
    ok = inst.reset(w, h, opt)
    if not ok:
        log(f"  reset failed for {w}x{h} d=[{dmin},{dmax})")
        return False
    t0 = time.time()
    res = inst.match(l, r)
    t_gpu = time.time() - t0
    if res is None:
        log("  match returned false")
        return False
    got = inst.read_stages()
    all_ok = True
    line = []
    for n in STAGE_NAMES:
        if n == "disp_r" and not opt.is_check_lr:
            continue
        same = feq(got[n], ref[n])
        line.append(f"{n}:{'ok' if same else 'DIFF'}")
        if not same:
            all_ok = False
            bad = np.argwhere(got[n] != ref[n]) if got[n].dtype != np.float32 else np.argwhere(got[n].view(np.uint32) != ref[n].view(np.uint32))
            log(f"    {n}: {len(bad)} mismatches of {got[n].size}; first {bad[:6].tolist()}")
            for idx in bad[:4]:
                idx = tuple(idx)
                log(f"      at {idx}: gpu={got[n][idx]} cpu={ref[n][idx]}")
    same = feq(res, ref["final"])
    log(f"  {w}x{h} d=[{dmin},{dmax}) {kw} -> {' '.join(line)} result:{'ok' if same else 'DIFF'}  cpu {t_cpu*1e3:.0f} ms gpu(call) {t_gpu*1e3:.1f} ms")
    if not (got["aggr"] == ref["aggr"]).all():
        # localise: per-direction planes vs the oracle's per-direction last-visit costs
        for i, (dx, dy) in enumerate(DIRECTIONS):
            _, last, vis = orc.aggregate_dir(l, ref["cost"], opt.p1, opt.p2_init, dx, dy, want_last=True, want_visits=True)
            plane = inst.read_stage(10 + i)
            once = vis == 1
            bad = (plane != last).any(axis=2) & once
            log(f"    dir {i} ({dx},{dy}): pixels visited once with wrong L_r: {int(bad.sum())} of {int(once.sum())}; "
                f"first {np.argwhere(bad)[:5].tolist()}")
    return all_ok and same


def main():
    orc = Oracle()
    inst = S.SGMInstance(0)
    log("library:", S.load_library().SGM_Version())
    cases = [
        (24, 16, 0, 8, 1, {"min_speckle_area": 6}),
        (70, 33, 0, 16, 2, {"min_speckle_area": 12}),
        (64, 20, 0, 40, 3, {"min_speckle_area": 10, "p1": 7, "p2_init": 99}),
        (33, 33, 0, 12, 4, {"min_speckle_area": 6}),
        (40, 24, 3, 19, 5, {"min_speckle_area": 8}),
        (20, 31, 0, 8, 6, {"min_speckle_area": 6}),
        (96, 40, 0, 32, 7, {}),
        (200, 50, 0, 64, 8, {}),
        (300, 60, 0, 128, 9, {}),
        (300, 60, 0, 192, 10, {}),
        (400, 48, 0, 256, 11, {"min_speckle_area": 20}),
        (40, 1100, 0, 8, 14, {"min_speckle_area": 6}),
        (450, 375, 0, 64, 12, {}),
        (1242, 375, 0, 128, 13, {}),
    ]
    n_ok = 0
    for (w, h, dmin, dmax, seed, kw) in cases:
        try:
            n_ok += bool(run_case(inst, orc, w, h, dmin, dmax, 0x5EED0000 + seed, **kw))
        except Exception:
            log(traceback.format_exc())
    log(f"{n_ok}/{len(cases)} cases bit-exact")
    inst.enable_timing(True)
    l, r = orc.synth_pair(1242, 375, 128, 0x5EED0002)
    opt = default_option(128)
    inst.keep_stages(False)
    for it in range(3):
        inst.reset(1242, 375, opt)
        t0 = time.time()
        inst.match(l, r)
        log(f"KITTI host-call {1e3*(time.time()-t0):.2f} ms; stages {inst.last_timing()}")
    inst.close()


if __name__ == "__main__":
    main()
