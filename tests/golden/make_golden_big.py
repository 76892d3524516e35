#!/usr/bin/env python3
"""Digests of the two LARGE BASELINE.json shapes, produced by the REFERENCE ITSELF (minutes of CPU time, so
kept apart from make_golden.py):

    C3  Middlebury full-res  2880 x 1988, D = 256   seed 0x5EED0003
    C5  DrivingStereo        1762 x 800,  D = 192   seed 0x5EED0005

    oracle/build_ref.sh 2880 1988 256 && oracle/build_ref.sh 1762 800 192
    python tests/golden/make_golden_big.py          -> tests/golden/cases_big.json

Expected values come from oracle/_ref/libsgm_ref_<shape>.so (the reference's SemiGlobalMatching.c, guarded
build); our restatement only supplies the seeded input generator.  The GPU parity test compares the digests of
every stage it can read back at these sizes (tests/test_gpu_parity.py::test_full_size_digests)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.pyoracle import STAGE_NAMES, Oracle, Reference, default_option, sha  # noqa: E402
from make_golden import opt_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
BIG = [("c5_drivingstereo_1762x800_d192", 1762, 800, 192, 0x5EED0005),
       ("c3_middlebury_2880x1988_d256", 2880, 1988, 256, 0x5EED0003)]


def main():
    # the reference's RemoveSpeckles keeps a uint32[MAX_IMG_SIZE] list on the stack (SemiGlobalMatching.c:589):
    # 23 MB at 2880x1988, so the 8 MB default stack limit has to go (main thread: the limit is read on growth)
    import resource
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    gen = Oracle()
    cases = []
    for name, w, h, d, seed in BIG:
        ref = Reference.for_shape(w, h, d)
        assert ref is not None, f"oracle/build_ref.sh {w} {h} {d} first"
        left, right = gen.synth_pair(w, h, d, seed)
        opt = default_option(d)
        t0 = time.time()
        st = ref.run(left, right, opt)
        entry = {"name": name, "w": w, "h": h, "d": d, "seed": seed, "option": opt_dict(opt),
                 "oob_dropped": ref.oob_count(), "reference_seconds": round(time.time() - t0, 1),
                 "sha256": {n: sha(st[n]) for n in STAGE_NAMES},
                 "sha256_inputs": {"left": sha(left), "right": sha(right)},
                 "invalid_final": int(np.isinf(st["final"]).sum()),
                 "aggr_sum": int(st["aggr"].sum(dtype=np.uint64)), "aggr_max": int(st["aggr"].max())}
        cases.append(entry)
        print(name, entry["reference_seconds"], "s invalid", entry["invalid_final"], flush=True)
        del st
    with open(os.path.join(OUT, "cases_big.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_big.py", "cases": cases}, f, indent=1)


if __name__ == "__main__":
    main()
