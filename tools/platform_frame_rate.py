#!/usr/bin/env python3
"""Rate of the test platform's frame format through sgm_match_planes, on the GPU box.

The platform's frames are fixed 1280 x 720 (server.py:23-24, frame_buffer.h:6-7): six byte planes in (5.5 MB), one float32
depth map out (3.7 MB), grey conversion + SGM + disparity -> depth on the device.  Measured:
  blocking   sgm_reset + sgm_match_planes per frame, pageable numpy arrays (what sgm_board_client does per message, minus TCP)
  pipelined  3 instances x 4 frames per call (RATE_INSTANCES, RATE_BATCH), one host thread each, sgm_match_planes_async on pinned
             buffers (2 x 4: 1410 fps, 1 x 4: 1120, 4 x 3: 1570)
and every output of the last round is compared with  oracle(board grey) -> platform depth formula  bit for bit.

    python tools/platform_frame_rate.py [D] > gpurun_out/platform_frame_rate.json"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W, H = 1280, 720


def main():
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import Oracle, default_option
    from oracle.platform_oracle import board_gray, disparity_to_depth
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    budget = float(os.environ.get("RATE_SECONDS", "4"))
    fx, baseline, doffs = 1733.74, 536.62, 0.0
    opt, oopt = S.default_option(d), default_option(d)
    orc = Oracle()
    rng = np.random.default_rng(1)
    n_frames = 4
    frames, want = [], []
    for k in range(n_frames):
        l, r = orc.synth_pair(W, H, d, 0xB0A2D + k)
        p = np.empty((6, H, W), np.uint8)
        for v, g in enumerate((l, r)):
            for c in range(3):
                p[3 * v + c] = np.clip(g.astype(np.int32) + rng.integers(-6, 7, (H, W)), 0, 255)
        frames.append(p)
        disp = orc.run(board_gray(p[0], p[1], p[2]), board_gray(p[3], p[4], p[5]), oopt)["final"]
        want.append(disparity_to_depth(disp, fx, baseline, doffs))

    def same(got, k):
        w = want[k]
        ok = ~np.isnan(w)
        return bool(np.array_equal(np.isnan(got), np.isnan(w)) and np.array_equal(got[ok].view(np.uint32), w[ok].view(np.uint32)))

    res = {"frame": [W, H], "disparity_range": d, "bytes_in": 6 * W * H, "bytes_out": 4 * W * H}
    # blocking, one frame per call
    inst = S.SGMInstance(0)
    out = np.empty((H, W), np.float32)
    n, t_sum, good = 0, 0.0, True
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < budget:
        t0 = time.perf_counter()
        assert inst.reset(W, H, opt) and inst.match_planes(frames[n % n_frames], fx, baseline, doffs, out)
        if n >= 2:
            t_sum += time.perf_counter() - t0
        good = good and (n >= n_frames or same(out, n % n_frames))
        n += 1
    inst.close()
    res["blocking"] = {"ms_per_frame": round(t_sum / (n - 2) * 1e3, 3), "fps": round((n - 2) / t_sum, 1), "frames": n - 2, "verified": good,
                       "entry": "sgm_reset + sgm_match_planes, pageable arrays"}
    # pipelined
    B, n_inst = int(os.environ.get("RATE_BATCH", "4")), int(os.environ.get("RATE_INSTANCES", "3"))
    insts = [S.SGMInstance(0, batch=B) for _ in range(n_inst)]
    bufs = []
    for k, i in enumerate(insts):
        assert i.set_overlap_post(True) and i.reset(W, H, opt)
        P, O = i.host_array((B, 6, H, W), np.uint8), i.host_array((B, H, W), np.float32)
        for j in range(B):
            P[j] = frames[(k * B + j) % n_frames]
        bufs.append((P, O))
        assert i.match_planes(P, fx, baseline, doffs, O)
    rounds, fail, stop_at = [0] * n_inst, [], [0.0]

    def worker(k):
        i, (P, O) = insts[k], bufs[k]
        while time.perf_counter() < stop_at[0]:
            if not (i.reset(W, H, opt) and i.match_planes(P, fx, baseline, doffs, O, wait=False) and i.match_wait()):
                fail.append(k)
                return
            rounds[k] += 1

    t0 = time.perf_counter()
    stop_at[0] = t0 + budget
    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_inst)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    ver = not fail and all(same(bufs[k][1][j], (k * B + j) % n_frames) for k in range(n_inst) for j in range(B))
    for i in insts:
        i.close()
    fps = sum(rounds) * B / el
    res["pipelined_pinned"] = {"fps": round(fps, 1), "frames": sum(rounds) * B, "instances": n_inst, "frames_per_call": B, "verified": ver,
                               "pcie_gb_per_s": round(fps * 10 * W * H / 1e9, 2),
                               "entry": "sgm_reset + sgm_match_planes_async + sgm_match_wait, sgm_host_alloc buffers, post pass on the second stream"}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
