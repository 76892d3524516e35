#!/usr/bin/env python3
"""Collect the per-kernel hardware counters bench.py's `roofline` uses -- run ON THE GPU BOX:

    cd /tmp && export TMPDIR=/tmp && python3 $GRAFT_REPO_ROOT/tools/profile_counters.py [--workload W] [--out profiles/counters.json]

One rocprofv3 run per counter set (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc runs carry --kernel-trace
only), each wrapping `python3 tools/pmc_workload.py`:

  * HBM traffic (FETCH_SIZE, WRITE_SIZE): one frame per launch with the batch kernels' lane configuration forced
    (SGM_LANES_PER_PIXEL=8, SGM_HL=0, SGM_SUM_SEGMENTS=1: whole rows in the fused sum kernel) -- with 8 frames per launch these two counters under-read on this ROCm build
    (~1/4 of the bytes, round 1), so traffic is taken per frame and scales with the frames of a launch.  FETCH_SIZE is
    doubled (gfx950 tallies 128-byte read requests as 64 bytes, /opt/skills/guides/MI355X_MICROARCH.md, HBM); both are KiB.
  * VALU issue (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_INST_ANY, SQ_WAIT_ANY): 8
    frames per launch, the configuration bench.py times; divided by 8 -> per frame.
  * LDS (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE), same run shape.

The result is stamped with bench.source_id() (sha256 over csrc/*.hip, *.hpp): bench.py prints `traffic_stale: true`
when the kernels have changed since.  Raw per-pass CSVs stay under gpurun_out/counters/."""
import argparse
import csv
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, source_id  # noqa: E402

PASSES = [
    ("fetch", ["FETCH_SIZE"], 1, {"SGM_LANES_PER_PIXEL": "8", "SGM_HL": "0", "SGM_SUM_SEGMENTS": "1"}),
    ("write", ["WRITE_SIZE"], 1, {"SGM_LANES_PER_PIXEL": "8", "SGM_HL": "0", "SGM_SUM_SEGMENTS": "1"}),
    ("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVES"], 8, {}),
    ("lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"], 8, {}),
    ("grbm", ["GRBM_GUI_ACTIVE"], 8, {}),
]


def per_kernel(path):
    """{kernel base name: {counter: mean per dispatch}} (templates folded onto the base name; warm-up dispatch dropped)."""
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = re.split(r"[<(]", name)[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0] for c, v in cs.items()} for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="kitti_1242x375_d128_p8", choices=sorted(WORKLOADS))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "counters.json"),
                    help="written under gpurun_out/ (the only directory that travels back from the GPU box); copy it to profiles/counters.json")
    ap.add_argument("--scratch", default=os.path.join(ROOT, "gpurun_out", "counters"))
    args = ap.parse_args()
    kernels = defaultdict(dict)
    raw = {}
    big = WORKLOADS[args.workload][0] * WORKLOADS[args.workload][1] * WORKLOADS[args.workload][2] > 100e6
    for tag, ctrs, batch, env in PASSES:
        if big and batch > 2:
            batch = 2                  # the large shapes run 2 frames per launch in bench.py (a frame's planes are 4.6 - 11.7 GB)
        if big and batch == 1:
            env = dict(env, SGM_XCD_STRIPS="2")     # ... with the XCD-aware block numbering, which one frame per launch does not use by itself
        d = os.path.join(args.scratch, f"{args.workload}_{tag}")
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--pmc", *ctrs, "--kernel-trace", "-d", d, "-o", tag, "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), "--workload", args.workload, "--batch", str(batch)]
        e = dict(os.environ, **env)
        print("+", " ".join(cmd), flush=True)
        with open(os.path.join(d, "log.txt"), "w") as log:
            rc = subprocess.call(cmd, env=e, stdout=log, stderr=subprocess.STDOUT, timeout=600)
        found = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
        if rc != 0 or not found:
            print(f"  pass {tag} failed (rc {rc}); see {d}/log.txt", flush=True)
            continue
        pk = per_kernel(found[0])
        raw[tag] = {"frames_per_launch": batch, "env": env, "kernels": pk}
        for k, cs in pk.items():
            for c, v in cs.items():
                kernels[k][c + "_per_frame"] = v / batch
    out_k = {}
    for k, cs in sorted(kernels.items()):
        e = {}
        if "FETCH_SIZE_per_frame" in cs and "WRITE_SIZE_per_frame" in cs:
            rd = cs["FETCH_SIZE_per_frame"] * 1024 * 2          # KiB -> bytes, x2 gfx950 correction
            wr = cs["WRITE_SIZE_per_frame"] * 1024
            e.update({"read_bytes_per_frame": int(rd), "write_bytes_per_frame": int(wr), "hbm_bytes_per_frame": int(rd + wr)})
        if "SQ_INSTS_VALU_per_frame" in cs:
            e["valu_insts_per_frame"] = int(cs["SQ_INSTS_VALU_per_frame"])
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVES",
                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU",
                  "GRBM_GUI_ACTIVE"):
            if c + "_per_frame" in cs:
                e[c + "_per_frame"] = round(cs[c + "_per_frame"], 1)
        out_k[k] = e
    doc = {"generator": "tools/profile_counters.py", "source_id": source_id(), "workloads": {}}
    if os.path.exists(args.out):
        with open(args.out) as f:
            old = json.load(f)
        if old.get("source_id") == doc["source_id"]:
            doc["workloads"] = old.get("workloads", {})
    doc["workloads"][args.workload] = {
        "note": "rocprofv3 --pmc passes of tools/pmc_workload.py; traffic at 1 frame per launch (FETCH_SIZE x2, KiB), SQ counters "
                f"at {2 if big else 8} frames per launch / {2 if big else 8}; mean per dispatch without the first (warm-up) one",
        "kernels": out_k}
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    with open(os.path.join(args.scratch, f"{args.workload}_raw.json"), "w") as f:
        json.dump(raw, f, indent=1)
    print(json.dumps({k: v for k, v in out_k.items() if k.startswith("sgm_aggregate") or k.startswith("sgm_sum")}, indent=1))


if __name__ == "__main__":
    main()
