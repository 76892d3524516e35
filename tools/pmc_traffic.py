#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT/fetch -o f --output-format csv -- python3 bench.py --batch 1 --in-flight 1 ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT/write -o w --output-format csv -- python3 bench.py --batch 1 --in-flight 1 ...
    python tools/pmc_traffic.py OUT/fetch/f_counter_collection.csv OUT/write/w_counter_collection.csv

Both counters are in KiB... (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes); FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests as 64 bytes).  Values are the
mean per dispatch of each kernel (templates folded onto the kernel's base name), i.e. per frame when the run
used one frame per launch."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = re.split(r"[<(]", name)[0]
            acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = fetch.get(k, 0.0) * 1024 * 2          # KiB -> bytes, gfx950 correction x2
        wr = write.get(k, 0.0) * 1024
        out[k] = {"fetch_size_kb_raw": round(fetch.get(k, 0.0), 1), "read_bytes_corrected": int(rd),
                  "write_bytes": int(wr), "hbm_bytes": int(rd + wr)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
