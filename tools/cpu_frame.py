#!/usr/bin/env python3
"""cpu_frame.py W H D SEED -- one warm-up frame and one timed frame of the synthetic workload through the reference's own C
(oracle/_ref, or our CPU restatement where that is absent); prints the seconds of the timed frame.  bench.py's
`cpu_baseline.all_cores` leg starts one of these per host core.  Checker-side code: never part of the product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Reference, default_option  # noqa: E402

w, h, d, seed = (int(v) for v in sys.argv[1:5])
opt = default_option(d)
ref = Reference.for_shape(w, h, d)
left, right = Oracle().synth_pair(w, h, d, seed)
if ref is not None:
    def run():
        return ref.api_match(left, right, opt, reset=True)
else:
    orc = Oracle()

    def run():
        assert orc.reset(w, h, opt)
        return orc.match(left, right)
assert run() is not None                    # first touch of the static buffers
t0 = time.perf_counter()
assert run() is not None
print(time.perf_counter() - t0)
