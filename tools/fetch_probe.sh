#!/usr/bin/env bash
# fetch_probe.sh WORKLOAD -- FETCH_SIZE of the aggregation kernel with and without the XCD-aware block numbering (SGM_XCD_STRIPS),
# 1 and 2 frames per launch.  On the GPU box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
wl=${1:-middlebury_2880x1988_d256_p8}
cd /tmp && export TMPDIR=/tmp
for b in 1 2; do for s in 0 1; do
  d=$R/gpurun_out/fetch_probe/${wl}_b${b}_s${s}; rm -rf $d; mkdir -p $d
  SGM_XCD_STRIPS=$s rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $d -o f --output-format csv -- python3 $R/tools/pmc_workload.py --workload $wl --batch $b --steps 3 > $d/log.txt 2>&1
  python3 - $d $b $s <<'PY'
import csv, glob, sys
d, b, s = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "sgm_aggregate_k" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"batch {b} strips {s}: aggregation FETCH_SIZE x2 = {sum(v[1:]) / len(v[1:]) * 2048 / 1e9 / int(b):.3f} GB per frame ({len(v)} launches)")
PY
done; done
