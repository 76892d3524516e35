#!/usr/bin/env python3
"""After `gpurun -- 'bash tools/refresh_profiles.sh'`: copy what that run left under gpurun_out/ into profiles/ as round N's files
(counters.json, rNN_bench.json = stdout, rNN_bench_detail.json, kernel trace summary, kernel stats) and rewrite the numbers
paragraph of profiles/README.md from them.  Refuses files measured on other kernel sources.   python tools/adopt_profiles.py 04"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

r = sys.argv[1] if len(sys.argv) > 1 else "04"
g, p = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
c = json.load(open(os.path.join(g, "counters.json")))
d = json.load(open(os.path.join(g, "bench_detail.json")))
line = open(os.path.join(g, "bench.json")).read()
assert c["source_id"] == bench.source_id() == d["source_id"], (c["source_id"], d["source_id"], bench.source_id())
assert d["frames_mismatched"] == 0 and json.loads(line) == bench.compact_line(d)
shutil.copy(os.path.join(g, "counters.json"), os.path.join(p, "counters.json"))
shutil.copy(os.path.join(g, "bench_detail.json"), os.path.join(p, f"r{r}_bench_detail.json"))
open(os.path.join(p, f"r{r}_bench.json"), "w").write(line)
shutil.copy(os.path.join(g, "bench_under_rocprof.json"), os.path.join(p, f"r{r}_bench_under_rocprof.json"))
shutil.copy(os.path.join(g, "kernel_trace_summary.json"), os.path.join(p, f"r{r}_kernel_trace_summary.json"))
shutil.copy(os.path.join(g, "prof_stats", "stats_kernel_stats.csv"), os.path.join(p, f"r{r}_kernel_stats.csv"))
rp = os.path.join(p, "README.md")
s = open(rp).read()
new = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "profiles_numbers.py"), r], text=True).strip()
head = f"Numbers of these files (`python tools/profiles_numbers.py {r}` prints them from the files; do not edit by hand): "
if head in s:
    i0 = s.index(head)
    i1 = s.index("## Round", i0)
    s = s[:i0] + head + new + "\n\n" + s[i1:]
s = re.sub(r"\(`source_id` [0-9a-f]{16}\)", f"(`source_id` {d['source_id']})", s, count=1)
s = re.sub(r"ONE compact line, \d+ bytes", f"ONE compact line, {len(line)} bytes", s, count=1)
open(rp, "w").write(s)
print(f"round {r}: {d['fps']:.0f} fps, source_id {d['source_id']}, line {len(line)} bytes")
