#!/usr/bin/env python3
"""Digests of EVERY frame bench.py times, produced by the REFERENCE ITSELF.

bench.py streams, per rank, two batches of B synthetic pairs with seeds  base + k*B + j  (k = batch, j = frame;
rank 0) and hashes every frame of the last timed batch of each in-flight instance against the digests written
here (`frames_verified` in the bench line).  tests/test_gpu_parity.py::test_batched_fast_path_* use the same file.

    oracle/build_ref.sh 1242 375 128 && oracle/build_ref.sh 450 375 64 && oracle/build_ref.sh 1762 800 192
    oracle/build_ref.sh 2880 1988 256 && oracle/build_ref.sh 3840 2160 256
    python tests/golden/make_golden_bench.py [workload ...]      -> tests/golden/bench_frames.json (merged)

Expected values come from oracle/_ref/libsgm_ref_<shape>.so (the reference's SemiGlobalMatching.c, guarded build,
SURVEY.md 8c); our restatement only supplies the seeded input generator.  One process per frame (the reference
keeps its state in globals), a few at a time (the 4K D=256 frame needs 6.4 GB of static buffers)."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_frames.json")

# name: (W, H, D, first seed, frame indices)   -- the names and seeds of bench.py's WORKLOADS; frame f has seed first + f
WORKLOADS = {
    "kitti_1242x375_d128_p8": (1242, 375, 128, 0x5EED0002, range(32)),          # BASELINE config 4: a batch of 32 frames
    "cone_450x375_d64_p8": (450, 375, 64, 0x5EED0001, range(16)),
    # BASELINE config 5 is a stream: bench.py pushes 256 distinct frames through and checks the first and the last four
    # (and its device-resident leg runs batches of 8: the first 16 frames)
    "drivingstereo_1762x800_d192_p8": (1762, 800, 192, 0x5EED0005, list(range(16)) + list(range(252, 256))),
    "middlebury_2880x1988_d256_p8": (2880, 1988, 256, 0x5EED0003, range(4)),
    "uhd_3840x2160_d128_p8": (3840, 2160, 128, 0x5EED0006, range(8)),
    "uhd_3840x2160_d256_p8": (3840, 2160, 256, 0x5EED0007, range(4)),
    # BASELINE config 1 says "4 paths": not a mode the reference can be asked for (num_paths is never read).  The mode is
    # DEFINED as the first four of the reference's eight CostAggregate calls (SemiGlobalMatching.c:213-216); these digests come from
    # the reference's own stage functions run with exactly those four calls (oracle/ref_harness_tail.c: ref_run_stages_first_dirs).
    "cone_450x375_d64_p4": (450, 375, 64, 0x5EED0001, range(16)),
    # throughput is reported with speckle removal on AND off; the same KITTI frames with is_remove_speckles = false
    # (main.c:60 flipped), so that the speckle-off rate is a verified one too
    "kitti_1242x375_d128_p8_nospeckle": (1242, 375, 128, 0x5EED0002, range(16)),
}
FIRST_FOUR_CALLS = {"cone_450x375_d64_p4"}
SPECKLE_OFF = {"kitti_1242x375_d128_p8_nospeckle"}
KEEP = ["disp_l", "disp_r", "after_lr", "after_speckle", "final"]


def one_frame(job):
    name, w, h, d, seed = job
    import resource
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))   # SemiGlobalMatching.c:588-589
    import numpy as np
    from oracle.pyoracle import Oracle, Reference, default_option, sha
    ref = Reference.for_shape(w, h, d)
    assert ref is not None, f"oracle/build_ref.sh {w} {h} {d} first"
    left, right = Oracle().synth_pair(w, h, d, seed)
    t0 = time.time()
    if name in FIRST_FOUR_CALLS:
        st = ref.run(left, right, default_option(d, num_paths=4), first_dirs=4)
    else:
        st = ref.run(left, right, default_option(d, is_remove_speckles=False) if name in SPECKLE_OFF else default_option(d))
    entry = {"sha256": {n: sha(st[n]) for n in KEEP}, "sha256_inputs": {"left": sha(left), "right": sha(right)},
             "invalid_final": int(np.isinf(st["final"]).sum()), "oob_dropped": ref.oob_count(),
             "reference_seconds": round(time.time() - t0, 1)}
    if name in FIRST_FOUR_CALLS:
        entry["made_by"] = "the reference's stage functions with the first four CostAggregate calls (ref_run_stages_first_dirs)"
    return name, seed, entry


def main():
    force = "--force" in sys.argv[1:]
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(WORKLOADS)
    doc = {"generator": "tests/golden/make_golden_bench.py", "workloads": {}}
    if os.path.exists(OUT):
        with open(OUT) as f:
            doc = json.load(f)
    jobs = []
    for n in names:
        w, h, d, seed, frames = WORKLOADS[n]
        old = {} if force else doc["workloads"].get(n, {}).get("frames", {})
        doc["workloads"][n] = {"w": w, "h": h, "d": d, "first_seed": seed,
                               "option": "main.c:48-65 with max_disparity = D" + (", num_paths = 4 honoured = the reference's first four CostAggregate calls" if n in FIRST_FOUR_CALLS else "")
                                         + (", is_remove_speckles = false" if n in SPECKLE_OFF else ""),
                               "frames": dict(old)}
        jobs += [(n, w, h, d, seed + k) for k in frames if str(seed + k) not in old]     # only what is missing (--force: all)
    jobs.sort(key=lambda j: -(j[1] * j[2] * j[3]))            # big frames first
    procs = int(os.environ.get("GOLDEN_PROCS", "3"))
    with mp.get_context("spawn").Pool(procs, maxtasksperchild=1) as pool:
        for name, seed, entry in pool.imap_unordered(one_frame, jobs):
            doc["workloads"][name]["frames"][str(seed)] = entry
            print(name, hex(seed), entry["reference_seconds"], "s", flush=True)
    for n in names:
        fr = doc["workloads"][n]["frames"]
        doc["workloads"][n]["frames"] = {k: fr[k] for k in sorted(fr, key=int)}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
