#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: per-kernel mean duration, and for the steady-state middle of the run
how much wall time had N kernels active and which kernels were active alone."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sgm" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0], r["Queue_Id"]) for r in rows)
n = len(ev)
lo, hi = ev[n // 4][0], ev[3 * n // 4][0]
pts = []
for s, e, k, q in ev:
    if e < lo or s > hi: continue
    pts.append((max(s, lo), 1, k)); pts.append((min(e, hi), -1, k))
pts.sort()
active = collections.Counter(); hist = collections.Counter(); alone = collections.Counter(); share = collections.Counter()
last = lo
for t, dlt, k in pts:
    nact = sum(active.values())
    hist[nact] += t - last
    for kk, c in active.items():
        if c: share[kk] += (t - last) * c / max(nact, 1)
    if nact == 1:
        alone[[kk for kk, c in active.items() if c][0]] += t - last
    active[k] += dlt
    last = t
tot = hi - lo
frames = sum(1 for s, e, k, q in ev if k == "sgm_census_k" and lo <= s < hi)
print(f"window {tot/1e6:.2f} ms, {frames} frames -> {tot/1e6/max(frames,1):.3f} ms/frame; queues {sorted(set(q for *_, q in ev))}")
print("time with N kernels active:", {k: f"{100*v/tot:.0f}%" for k, v in sorted(hist.items())})
print("per-frame 'fair share' time by kernel (us):", {k: round(v / 1e3 / max(frames, 1), 1) for k, v in share.most_common()})
print("time alone on the GPU per frame (us):", {k: round(v / 1e3 / max(frames, 1), 1) for k, v in alone.most_common()})
dur = collections.defaultdict(list)
for s, e, k, q in ev:
    if lo <= s < hi: dur[k].append(e - s)
print("mean duration (us):", {k: round(sum(v) / len(v) / 1e3, 1) for k, v in dur.items()})
