#!/usr/bin/env python3
"""trace_summary.py KERNEL_TRACE.csv BENCH_LINE.json -- per kernel of a `rocprofv3 --kernel-trace` run of bench.py: calls, mean
duration over all dispatches and over the LAST `timed` ones (timed = launches_timed of the bench line = the dispatches of the timed
region), next to the bench line's own HIP-event means for the two heavy kernels."""
import csv
import json
import re
import sys
from collections import defaultdict

trace, line = sys.argv[1], json.load(open(sys.argv[2]))
timed = line["roofline"]["launches_timed"]
per = defaultdict(list)
for r in csv.DictReader(open(trace)):
    name = re.split(r"[<(]", re.sub(r"^void ", "", r["Kernel_Name"]))[0]
    per[name].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
out = {"timed_launches_per_kernel": timed, "kernels": {}}
for k, v in sorted(per.items()):
    d = [x[1] for x in sorted(v)]
    out["kernels"][k] = {"calls": len(d), "mean_ms_all": round(sum(d) / len(d), 4), "mean_ms_timed": round(sum(d[-timed:]) / len(d[-timed:]), 4)}
out["bench_line_event_means_ms"] = {"sgm_aggregate_k": line["roofline"]["avg_launch_ms"], "sgm_sum_wta_lr_k": line["roofline_sum_wta"]["avg_launch_ms"]}
print(json.dumps(out, indent=1))
