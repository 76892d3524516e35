"""The C stream driver (csrc/sgm_stream.c -> soc_project_stereo_matching_amd/sgm_stream): a C caller with threads, instances and
page-locked buffers -- no Python in the measured process.  Its disparity map of frame 0 (FNV-1a of the float bytes) must be the
oracle's, in every mode; the rates it prints are what profiles/ quotes for "the C path"."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "soc_project_stereo_matching_amd", "sgm_stream")


def fnv1a(b: bytes) -> str:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.mark.parametrize("mode", [[], ["--pageable"], ["--blocking"], ["--batch", "1", "--instances", "2"]],
                         ids=["pinned_batches", "pageable_batches", "blocking", "single_frames"])
def test_c_stream_driver_matches_the_oracle(oracle, mode):
    from oracle.pyoracle import default_option
    if not os.path.exists(EXE):
        pytest.skip("sgm_stream not built")
    w, h, d, seed = 320, 96, 64, 77
    left, right = oracle.synth_pair(w, h, d, seed)
    want = oracle.run(left, right, default_option(d))["final"]
    out = subprocess.run([EXE, "--width", str(w), "--height", str(h), "--disparities", str(d), "--seed", str(seed), "--frames", "12",
                          "--seconds", "0.5"] + mode, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-800:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["hash_frame0"] == fnv1a(np.ascontiguousarray(want).tobytes())
    assert line["frames"] > 10 and line["fps"] > 100 and not line.get("failed", False)
