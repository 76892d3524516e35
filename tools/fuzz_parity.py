#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (diagnostics; the committed parity tests are under tests/):
random shapes / ranges / options / modes (batch, fused or separate kernels, row tiles, extensions) against the CPU oracle,
every stage.  python tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import soc_project_stereo_matching_amd as S  # noqa: E402
from oracle.pyoracle import STAGE_NAMES, Oracle, default_option  # noqa: E402


def same(a, b):
    return np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)


def tiles_case(orc, rng, w, h, d, opt, frame, seed, kw):
    """one frame cut into 2..5 row tiles (instances standing in for GPUs), against the oracle's final map"""
    import torch
    from soc_project_stereo_matching_amd.tiling import DeviceTileEngine, match_tiled_in_process, tile_rows
    n_tiles = int(min(h, rng.integers(2, 6)))
    want = orc.run(frame[0], frame[1], opt)["final"]
    engines = []
    try:
        for rows in tile_rows(h, n_tiles):
            engines.append(DeviceTileEngine(0, w, h, opt, rows))
        got = match_tiled_in_process(engines, torch.from_numpy(frame[0]).cuda(), torch.from_numpy(frame[1]).cuda()).cpu().numpy()
        if not same(got, want):
            print(f"MISMATCH tiles {w}x{h} d={d} tiles={n_tiles} seed={seed} opts={kw}", flush=True)
            return 1
        return 0
    finally:
        for e in engines:
            e.close()


def pipe_case(orc, rng, w, h, d, opt, seed, kw):
    """the frames-in-flight tile pipeline: 2..4 ranks as threads of this process, batches of 1..3 frames per step, a lead of 0..2"""
    import threading
    import torch
    from soc_project_stereo_matching_amd.tiling import DeviceSlotEngine, InProcessGroup, TilePipeline, tile_rows
    ranks = int(min(h, rng.integers(2, 5)))
    B, steps, lead = int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(0, 3))
    frames = [[orc.synth_pair(w, h, d, seed + 8 * k + j) for j in range(B)] for k in range(steps)]
    if B > 1:
        dev = [(torch.from_numpy(np.stack([p[0] for p in fr])).cuda(), torch.from_numpy(np.stack([p[1] for p in fr])).cuda()) for fr in frames]
    else:
        dev = [(torch.from_numpy(fr[0][0]).cuda(), torch.from_numpy(fr[0][1]).cuda()) for fr in frames]
    torch.cuda.synchronize()
    group = InProcessGroup(ranks, timeout=60)
    got, errors = {}, []

    def rank_main(r):
        eng = None
        try:
            torch.cuda.set_device(0)
            eng = DeviceSlotEngine(0, w, h, opt, tile_rows(h, ranks)[r], TilePipeline.slots_needed(ranks, lead), host_staged=False, batch=B)

            def on_result(f, t, ev):
                ev.synchronize()
                got[f] = t.cpu().numpy().copy()

            TilePipeline(eng, r, ranks, h, dist=group.view(r), lead=lead).run(steps, lambda f: dev[f], on_result)
        except Exception as exc:                                    # noqa: BLE001
            import traceback
            errors.append((r, repr(exc), traceback.format_exc()[-600:]))
        finally:
            if eng is not None:
                eng.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    bad = bool(errors) or sorted(got) != list(range(steps))
    for k in range(steps):
        for j in range(B):
            if bad:
                break
            g = got[k][j] if B > 1 else got[k]
            bad = not same(g, orc.run(frames[k][j][0], frames[k][j][1], opt)["final"])
    if bad:
        print(f"MISMATCH pipeline {w}x{h} d={d} ranks={ranks} B={B} steps={steps} lead={lead} seed={seed} errors={errors} opts={kw}", flush=True)
    return int(bad)


def cpipe_case(orc, rng, w, h, d, opt, seed, kw):
    """the C host of the tile pipeline (include/sgm_tiles.h): 1..5 ranks as threads over its local transport, batches of 1..3 frames
    per step, a lead of 0..3, more frames than slots now and then; every frame against the oracle"""
    import threading
    import torch
    from soc_project_stereo_matching_amd import tiles
    ranks = int(min(h, rng.integers(1, 6)))
    B, steps, lead = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 4))
    frames = [[orc.synth_pair(w, h, d, seed + 8 * k + j) for j in range(B)] for k in range(steps)]
    dev = [(torch.from_numpy(np.stack([p[0] for p in fr])).cuda(), torch.from_numpy(np.stack([p[1] for p in fr])).cuda()) for fr in frames]
    torch.cuda.synchronize()
    ring_frames = (steps + ranks - 1) // ranks
    rings = [torch.full((ring_frames, B, h, w), -3.0, dtype=torch.float32, device="cuda") for _ in range(ranks)]
    group = tiles.LocalGroup(ranks, 0) if ranks > 1 else None
    errors = []
    throttle = int(rng.integers(0, 4))                              # (the generator is not for the threads to share)

    def rank_main(r):
        try:
            tr = group.transport(r) if group else None
            pipe = tiles.TilesPipeline(0, r, ranks, w, h, opt, batch=B, lead=lead, throttle=throttle, transport=tr)
            pipe.result_ring(rings[r].data_ptr(), ring_frames)
            for k in range(steps):
                assert pipe.submit(dev[k][0].data_ptr(), dev[k][1].data_ptr()), f"submit {k}"
            assert pipe.finish(), "finish"
            pipe.close()
            if tr is not None:
                tr.close()
        except BaseException as exc:                                # noqa: BLE001
            errors.append((r, repr(exc)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    if group:
        group.close()
    bad = bool(errors)
    for k in range(steps):
        got = rings[k % ranks][(k // ranks) % ring_frames].cpu().numpy()
        for j in range(B):
            if bad:
                break
            bad = not same(got[j], orc.run(frames[k][j][0], frames[k][j][1], opt)["final"])
    if bad:
        print(f"MISMATCH cpipe {w}x{h} d={d} ranks={ranks} B={B} steps={steps} lead={lead} seed={seed} errors={errors} opts={kw}", flush=True)
    return int(bad)


def planes_case(orc, rng, w, h, d, opt, oopt, seed, kw):
    """a batch of test-platform frames (six colour planes each) through sgm_match_planes against oracle -> depth formula"""
    from oracle.platform_oracle import board_gray, disparity_to_depth
    B = int(rng.integers(1, 4))
    fx, baseline, doffs = float(rng.choice([1733.74, 3979.9, 100.0])), float(rng.choice([536.62, 193.0])), float(rng.choice([0.0, 124.3, -3.0]))
    planes = np.empty((B, 6, h, w), np.uint8)
    for j in range(B):
        l, r = orc.synth_pair(w, h, d, seed + j)
        for v, g in enumerate((l, r)):
            for c in range(3):
                planes[j, 3 * v + c] = np.clip(g.astype(np.int32) + rng.integers(-9, 10, (h, w)), 0, 255)
    inst = S.SGMInstance(0, batch=B)
    try:
        if rng.random() < 0.5:
            inst.set_overlap_post(True)
        pinned = rng.random() < 0.5
        src = inst.host_array(planes.shape, np.uint8) if pinned else np.empty_like(planes)
        src[...] = planes
        out = inst.host_array((B, h, w), np.float32) if pinned else np.empty((B, h, w), np.float32)
        assert inst.reset(w, h, opt), "reset"
        assert inst.match_planes(src if B > 1 else src[0], fx, baseline, doffs, out if B > 1 else out[0]), "match_planes"
        for j in range(B):
            disp = orc.run(board_gray(planes[j, 0], planes[j, 1], planes[j, 2]), board_gray(planes[j, 3], planes[j, 4], planes[j, 5]), oopt)["final"]
            want = disparity_to_depth(disp, fx, baseline, doffs)
            ok = ~np.isnan(want)
            if not (np.array_equal(np.isnan(out[j]), np.isnan(want)) and np.array_equal(out[j][ok].view(np.uint32), want[ok].view(np.uint32))):
                print(f"MISMATCH planes {w}x{h} d={d} B={B} frame {j} seed={seed} calib=({fx},{baseline},{doffs}) opts={kw}", flush=True)
                return 1
        return 0
    finally:
        inst.close()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    orc = Oracle()
    t0 = time.time()
    n = bad = 0
    while time.time() - t0 < budget:
        w = int(rng.choice([rng.integers(6, 80), rng.integers(80, 700), rng.integers(700, 1400)], p=[0.3, 0.55, 0.15]))
        h = int(rng.choice([rng.integers(6, 60), rng.integers(60, 400), rng.integers(514, 900)], p=[0.45, 0.45, 0.1]))
        d = int(rng.choice([1, 3, 8, 16, 24, 40, 64, 100, 128, 160, 192, 256, 300]))
        if w * h * d > 40e6:
            continue
        dmin = int(rng.choice([0, 0, 0, 2, 7]))
        kw = dict(min_speckle_area=int(rng.choice([1, 9, 50, 300])), is_check_lr=bool(rng.random() < 0.8),
                  is_check_unique=bool(rng.random() < 0.8), is_remove_speckles=bool(rng.random() < 0.8),
                  p1=int(rng.choice([10, 0, 3, 40, 120, 32767, -7, 255])), p2_init=int(rng.choice([150, 0, 20, 90, 400, 32767, -30, 256])),
                  uniqueness_ratio=float(rng.choice([0.99, 0.95, 0.8])), lrcheck_thres=float(rng.choice([1.0, 0.0, 2.5])))
        opt = default_option(dmin + d, dmin, **kw)
        seed = int(rng.integers(1, 2**31))
        modes = os.environ.get("FUZZ_MODES", "plain,plain,batch,batch,separate,window,rightview,tiles,planes,pipe,cpipe,upsum").split(",")
        mode = str(rng.choice(modes))
        if mode == "upsum":
            # the fused last sweep (SGM_UPSUM=1, csrc/sgm_upsum.hip): it needs W > H and a padded range of 128 -- make most draws qualify
            # (the others, and negative P1, fall back to the separate kernels inside the library: still a valid case)
            if rng.random() < 0.85:
                d = int(rng.choice([128, 128, 100, 117, 70, 65]))
                w = max(w, 64)
                h = int(min(h, w - 1, rng.integers(4, 200)))
                opt = default_option(dmin + d, dmin, **kw)
            if w * h * d > 40e6:
                continue
        if mode == "tiles" and h < 4:
            mode = "plain"
        B = int(rng.integers(2, 5)) if mode == "batch" else (int(rng.integers(1, 4)) if mode == "upsum" else 1)
        frames = [orc.synth_pair(w, h, d, seed + k) for k in range(B)]
        if mode == "separate":
            os.environ["SGM_FUSED_WTA"] = "0"
        if mode == "upsum":                                   # read when the instance is created
            os.environ["SGM_UPSUM"] = "1"
            os.environ["SGM_UPSUM_ROWS"] = str(int(rng.integers(1, 4)))
            os.environ["SGM_UPSUM_WGS"] = str(int(rng.choice([1, 2, 5, 16, 48])))
        win = (5, 5)
        if mode == "window":
            win = [(7, 7), (9, 7), (3, 5), (7, 9), (1, 1), (63, 1)][int(rng.integers(0, 6))]
        orc.set_census_window(*win)
        orc.set_reference_view(mode == "rightview")
        if mode == "cpipe" and h >= 5:
            n += 1
            bad += cpipe_case(orc, rng, w, h, d, opt, seed, kw)
            continue
        if mode == "pipe" and h >= 4:
            n += 1
            bad += pipe_case(orc, rng, w, h, d, opt, seed, kw)
            continue
        if mode == "planes":
            n += 1
            bad += planes_case(orc, rng, w, h, d, S.default_option(dmin + d, dmin, **kw), opt, seed, kw)
            continue
        if mode == "tiles":
            n_bad = tiles_case(orc, rng, w, h, d, opt, frames[0], seed, kw)
            n += 1
            bad += n_bad
            continue
        inst = S.SGMInstance(0, batch=B)
        inst.set_census_window(*win)
        inst.set_reference_view(mode == "rightview")
        keep = bool(rng.random() < (0.15 if mode == "upsum" else 0.5))
        inst.keep_stages(keep)
        if rng.random() < 0.5:
            inst.set_overlap_post(True)                   # post pass on the instance's second stream
        try:
            assert inst.reset(w, h, opt), "reset"
            wants = [orc.run(l, r, opt) for l, r in frames]
            L = np.stack([f[0] for f in frames]) if B > 1 else frames[0][0]
            R = np.stack([f[1] for f in frames]) if B > 1 else frames[0][1]
            out = inst.match(L, R)
            assert out is not None, "match"
            diffs = []
            for k in range(B):
                got_final = out[k] if B > 1 else out
                if not same(got_final, wants[k]["final"]):
                    diffs.append(f"frame {k} final")
                inst.select_frame(k)
                stages = STAGE_NAMES if keep else ["census_l", "census_r", "aggr", "final"] + (["disp_r"] if (opt.is_check_lr or mode == "rightview") else [])
                for name in stages:
                    if name == "disp_r" and not (opt.is_check_lr or mode == "rightview"):
                        continue
                    if not same(inst.read_stage(name), wants[k][name]):
                        diffs.append(f"frame {k} {name}")
            n += 1
            if mode == "upsum":
                n_up = globals().get("_n_up", [0, 0])
                n_up[0] += 1
                n_up[1] += 1 if inst.fused_sweep_rows() else 0
                globals()["_n_up"] = n_up
            if diffs:
                bad += 1
                print(f"MISMATCH {w}x{h} d=[{dmin},{dmin + d}) mode={mode} B={B} win={win} keep={keep} seed={seed} opts={kw}: {diffs[:6]}", flush=True)
        finally:
            inst.close()
            os.environ.pop("SGM_FUSED_WTA", None)
            for k_ in ("SGM_UPSUM", "SGM_UPSUM_ROWS", "SGM_UPSUM_WGS"):
                os.environ.pop(k_, None)
            orc.set_census_window(5, 5)
            orc.set_reference_view(False)
        if n % 20 == 0:
            print(f"{n} cases, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
    if "_n_up" in globals():
        print(f"upsum mode: {_n_up[0]} cases, {_n_up[1]} of them ran the fused last sweep")
    print(f"done: {n} cases, {bad} mismatching")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
