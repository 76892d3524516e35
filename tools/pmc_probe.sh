#!/usr/bin/env bash
# pmc_probe.sh "COUNTER COUNTER ..." [tag] -- one rocprofv3 --pmc pass of tools/pmc_workload.py (8 frames per launch) on the
# GPU box; prints the mean per dispatch of every counter for the aggregation and the fused sum kernel (diagnostics).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
tag=${2:-probe}
out=$R/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --kernel-trace -d "$out" -o p --output-format csv -- python3 $R/tools/pmc_workload.py --batch 8 > "$out/log.txt" 2>&1
python3 - "$out" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(f[0])):
    name = re.split(r"[<(]", re.sub(r"^void ", "", row["Kernel_Name"]))[0]
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in ("sgm_aggregate_k", "sgm_sum_wta_lr_k"):
    print(k, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in sorted(acc[k].items())})
PY
