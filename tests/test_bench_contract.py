"""bench.py without a GPU: what the driver's contract needs from it that can be checked on the CPU -- every workload the default
line times has reference digests for exactly the frames it will verify (tests/golden/bench_frames.json, made by the compiled
reference), the defaults finish in minutes, the roofline helper's arithmetic, the frame pool sharding."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_default_legs_have_reference_digests_for_every_frame_they_check():
    with open(os.path.join(ROOT, "tests", "golden", "bench_frames.json")) as f:
        golden = json.load(f)["workloads"]
    # headline: the 32-frame KITTI pool
    w, h, d, seed = bench.WORKLOADS["kitti_1242x375_d128_p8"]
    have = bench.golden_digests("kitti_1242x375_d128_p8")
    assert all(seed + f in have for f in range(bench.POOL_FRAMES))
    assert golden["kitti_1242x375_d128_p8"]["w"] == w and golden["kitti_1242x375_d128_p8"]["h"] == h
    # the other single-GPU configs of the line: (workload, frames per launch, batches verified)
    for name, B, n in (("cone_450x375_d64_p8", 8, 2), ("cone_450x375_d64_p4", 8, 2), ("middlebury_2880x1988_d256_p8", 2, 2),
                       ("drivingstereo_1762x800_d192_p8", 8, 2), ("kitti_1242x375_d128_p8_nospeckle", 8, 2)):
        seed = bench.WORKLOADS[name][3]
        have = bench.golden_digests(name)
        assert all(seed + f in have for f in range(n * B)), name
    # config 5 as a stream: first and last four of 256 frames
    seed = bench.WORKLOADS["drivingstereo_1762x800_d192_p8"][3]
    have = bench.golden_digests("drivingstereo_1762x800_d192_p8")
    assert all(seed + f in have for f in (0, 1, 2, 3, 252, 253, 254, 255))


def test_pool_is_sharded_frame_by_frame_over_the_ranks():
    from soc_project_stereo_matching_amd.sharding import frames_of_rank
    for world in (1, 2, 4, 8):
        shares = [frames_of_rank(bench.POOL_FRAMES, world, r) for r in range(world)]
        assert sorted(sum(shares, [])) == list(range(bench.POOL_FRAMES))
        assert all(len(s) == bench.POOL_FRAMES // world for s in shares)


def test_roofline_object_arithmetic():
    counters = {"source_id": bench.source_id(), "file": "x", "kernels": {"k": {"hbm_bytes_per_frame": 500e6, "valu_insts_per_frame": 50e6}}}
    r = bench.kernel_roofline("k", 1.0, 10, 0.9, 8, counters, 480e6, "note", ref_equiv_bytes_per_frame=2.4e9)
    assert r["traffic"] == 4_000_000_000 and r["achieved"] == 4000.0 and r["frac"] == 0.5 and not r["traffic_stale"]
    assert r["algorithmic_frac"] == round(480e6 * 8 / 1e-3 / 1e9 / 8000.0, 4)
    assert r["valu"]["frac"] == round(50e6 * 8 * 4 / (1024 * 2.4e9 * 1e-3), 4) and r["bound"] == "valu"
    assert r["reference_dataflow_equiv"]["x_peak"] == 2.4
    # counters of other kernel sources are reported but not used for the fraction
    stale = dict(counters, source_id="0000")
    r = bench.kernel_roofline("k", 1.0, 10, 0.9, 8, stale, 480e6, "note")
    assert r["traffic_stale"] and r["bytes_used"] == "algorithmic" and r["frac"] == r["algorithmic_frac"]


def test_committed_counters_belong_to_the_committed_kernels():
    """profiles/counters.json is stamped with the hash of csrc/*.hip, *.hpp it was measured on: a kernel change without a fresh
    counter run would silently turn the driver line's `traffic` into a stale number (bench.py then falls back to algorithmic bytes
    and says traffic_stale)."""
    with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
        doc = json.load(f)
    assert doc["source_id"] == bench.source_id()
    for wl in ("kitti_1242x375_d128_p8", "cone_450x375_d64_p8", "drivingstereo_1762x800_d192_p8", "middlebury_2880x1988_d256_p8"):
        k = doc["workloads"][wl]["kernels"]
        assert k["sgm_aggregate_k"]["hbm_bytes_per_frame"] > 0 and k["sgm_sum_wta_lr_k"]["valu_insts_per_frame"] > 0


def test_defaults_are_the_driver_contract():
    import argparse  # noqa: F401
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--gpus", type=int, default=1' in src and '"--steps"' in src and '"--warmup"' in src
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.PATHS == 8


def test_committed_bench_line_has_the_contract_keys():
    """profiles/r04_bench_detail.json is the full object behind the line the driver's command printed on an MI355X for the committed
    kernels (bench_detail.json of that run): the keys the contract names, both extra objects, every single-GPU BASELINE workload
    verified, nothing mismatched; profiles/r04_bench.json is the compact line itself as stdout carried it."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_detail.json")) as f:
        d = json.load(f)
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        raw = f.read()
    assert raw.count("\n") <= 1 and len(raw) < bench.COMPACT_LIMIT
    line = json.loads(raw)
    assert line == bench.compact_line(d) and line["value"] == d["value"] and line["roofline"]["frac"] == d["roofline"]["frac"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["source_id"] == bench.source_id() and d["n_gpus"] == 1 and d["config"]["workload"] == "kitti_1242x375_d128_p8"
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu") and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] and not r["traffic_stale"]
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] == 1
    assert d["frames_mismatched"] == 0 and d["frames_verified"] >= 32
    names = {w["workload"]: w for w in d["workloads"]}
    for wl in ("cone_450x375_d64_p8", "cone_450x375_d64_p4", "middlebury_2880x1988_d256_p8", "drivingstereo_1762x800_d192_p8",
               "kitti_1242x375_d128_p8_nospeckle"):
        assert names[wl]["frames_mismatched"] == 0 and names[wl]["frames_verified"] > 0 and names[wl]["roofline"], wl
    assert d["stream"]["frames"] >= 256 and d["stream"]["frames_mismatched"] == 0
    hb = d["host_boundary"]
    assert all(hb[k]["verified"] for k in ("blocking_single_frame", "blocking_single_frame_pinned", "pipelined_pageable"))



def _strict(o):
    """json.dumps(allow_nan=False): the line must be strict JSON (no NaN / Infinity tokens)."""
    return json.dumps(o, allow_nan=False)


@pytest.mark.parametrize("name", ["r03_bench.json", "r04_bench_detail.json"])
def test_compact_line_fits_a_tail_limited_reader(name):
    """BENCH_r03.parsed was null: bench.py printed one 21 KB object and the driver keeps only the tail of stdout.  The line rank 0
    prints (bench.compact_line of the full object) must stay below 4 KB, be strict JSON and carry the contract keys, `roofline`
    and `cpu_baseline` -- checked on full lines an MI355X produced."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not committed yet")
    with open(path) as f:
        full = json.load(f)
    if "all_cores" in (full.get("cpu_baseline") or {}):                 # round 3 spelling
        full["cpu_baseline"]["cores_16"] = full["cpu_baseline"].pop("all_cores")
    c = bench.compact_line(full)
    s = _strict(c)
    assert len(s) < bench.COMPACT_LIMIT == 4096, len(s)
    assert "\n" not in s and json.loads(s) == c
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus"):
        assert c[k] == full[k]
    assert c["config"]["workload"] == full["config"]["workload"] and "model" not in c["config"]
    r = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    cb = c["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert set(c["workloads"]) == {w["workload"] for w in full["workloads"]}
    for name_, row in c["workloads"].items():
        assert len(row) == len(c["workloads_columns"]) and row[2] == 0 and row[1] > 0, name_
    assert c["detail"] == "bench_detail.json"


def test_compact_line_survives_many_extra_legs():
    """Whatever the legs carry, the printed line stays below the limit (summaries are dropped before the contract keys)."""
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        full = json.load(f)
    full["workloads"] = [dict(w, workload=f"{w['workload']}_{k}") for k in range(40) for w in full["workloads"]]
    c = bench.compact_line(full)
    assert len(_strict(c)) < bench.COMPACT_LIMIT
    assert c["roofline"] and c["cpu_baseline"] and c["value"] == full["value"]


def test_bench_prints_exactly_one_stdout_line():
    """emit(): one stdout line (the compact one); the full object goes to stderr and bench_detail.json."""
    import contextlib
    import io
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        full = json.load(f)
    out, err = io.StringIO(), io.StringIO()
    detail = os.path.join(ROOT, "bench_detail.json")
    had = os.path.exists(detail)
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        bench.emit(full)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096 and json.loads(lines[0])["value"] == full["value"]
    assert json.loads(err.getvalue())["workloads"] == full["workloads"]
    with open(detail) as f:
        assert json.load(f)["value"] == full["value"]
    if not had:
        os.remove(detail)
