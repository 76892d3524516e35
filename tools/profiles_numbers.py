#!/usr/bin/env python3
"""Print the numbers paragraph of profiles/README.md for round N from the committed files themselves (so the README never quotes a
number its files do not hold):  python tools/profiles_numbers.py 04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = sys.argv[1] if len(sys.argv) > 1 else "04"
P = os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(P, f"r{r}_bench_detail.json")))
ts = json.load(open(os.path.join(P, f"r{r}_kernel_trace_summary.json")))
ur = json.load(open(os.path.join(P, f"r{r}_bench_under_rocprof.json")))
agg, sm = ts["kernels"]["sgm_aggregate_k"], ts["kernels"]["sgm_sum_wta_lr_k"]
al, sal = d["roofline"]["alone"], d["roofline_sum_wta"]["alone"]
wl = {w["workload"]: w for w in d["workloads"]}
hb = d["host_boundary"]
print(f"headline {d['fps']:.0f} fps through host pointers in the driver's 20 steps ({d['frames_verified']} / "
      f"{d['frames_verified'] + d['frames_mismatched']} frames = reference digests), {d['sustained']['fps']:.0f} in the 2-second loop, "
      f"{d['device_resident']['fps']:.0f} device-resident, single frame {d['single_frame_latency_ms']:.3f} ms, blocking sgm_compute "
      f"{hb['blocking_single_frame']['fps']:.0f} fps (page-locked {hb['blocking_single_frame_pinned']['fps']:.0f}), pageable pipelined "
      f"{hb['pipelined_pageable']['fps']:.0f} fps; aggregation {d['roofline']['avg_launch_ms']:.2f} ms per 8 frames in the timed pipeline "
      f"({d['roofline']['frac']:.2f} of the HBM peak by measured traffic, VALU {d['roofline']['valu']['frac']:.2f}), {al['avg_launch_ms']:.2f} ms alone "
      f"({al['frac']:.2f} / VALU {al['valu_frac']:.2f}); fused sum / WTA {d['roofline_sum_wta']['avg_launch_ms']:.2f} ms in flight "
      f"({d['roofline_sum_wta']['frac']:.2f}), {sal['avg_launch_ms']:.2f} ms alone ({sal['frac']:.2f}); kernel trace {agg['mean_ms_all']:.3f} / "
      f"{agg['mean_ms_timed']:.3f} ms (all / timed dispatches) against the event mean {ur['roofline']['avg_launch_ms']:.3f} for the aggregation, "
      f"{sm['mean_ms_all']:.3f} / {sm['mean_ms_timed']:.3f} against {ur['roofline_sum_wta']['avg_launch_ms']:.3f} for the sum; 2880x1988 D=256 "
      f"{wl['middlebury_2880x1988_d256_p8']['fps']:.0f} fps, 1762x800 D=192 {wl['drivingstereo_1762x800_d192_p8']['fps']:.0f} fps (stream "
      f"{d['stream']['fps']:.0f}), cone {wl['cone_450x375_d64_p8']['fps']:.0f} / 4 paths {wl['cone_450x375_d64_p4']['fps']:.0f} fps, KITTI without "
      f"speckle removal {wl['kitti_1242x375_d128_p8_nospeckle']['fps']:.0f} fps; reference C on one host core {d['cpu_baseline']['fps']:.3f} fps "
      f"({d['cpu_baseline']['host_cpu']}), 16 cores {d['cpu_baseline']['cores_16']['fps']:.1f} fps.")
