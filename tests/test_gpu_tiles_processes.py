"""The C tile pipeline (include/sgm_tiles.h) with every rank a PROCESS of its own: tests/tiles_rank.c (plain C: its own HIP
context, its own copy of libsgm_mi355x.so, no Python) x WORLD, connected by tests/sock_transport.c (Unix-domain sockets, staged
through the host -- RCCL refuses two ranks on one GPU, so this is how the multi-process path can be rehearsed on a one-GPU box).
Every frame of the stream, gathered on its owner rank, must be the oracle's map bit for bit.  What this does NOT cover is RCCL over
xGMI between GPUs (tests/test_gpu_tiling.py::test_c_rccl_transport_on_one_gpu exercises the RCCL transport with one rank)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, "soc_project_stereo_matching_amd")


@pytest.fixture(scope="module")
def rank_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tilesrank") / "tiles_rank")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-w", "-o", out, os.path.join(ROOT, "tests", "tiles_rank.c"),
                           os.path.join(ROOT, "tests", "sock_transport.c"), "-L", PKG, "-lsgm_mi355x", f"-Wl,-rpath,{PKG}", "-lm"])
    return out


@pytest.mark.parametrize("case", [
    # W, H, D, ranks, batch, lead, steps
    (300, 70, 48, 2, 1, 2, 5),
    (160, 50, 32, 3, 2, 1, 7),
    (1242, 375, 128, 4, 1, 2, 6),                  # KITTI size in 4 tiles
], ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}_n{c[3]}_b{c[4]}_lead{c[5]}")
def test_c_pipeline_ranks_as_processes(oracle, rank_exe, tmp_path, case):
    _run_ranks(oracle, rank_exe, tmp_path, case, None)


@pytest.mark.parametrize("case", [(300, 70, 48, 2, 2, 2, 5), (1242, 375, 128, 4, 8, 2, 6)],
                         ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}_n{c[3]}_b{c[4]}_lead{c[5]}")
def test_c_pipeline_over_rccl_between_gpus(oracle, rank_exe, tmp_path, case):
    """The same ranks on a GPU EACH, connected by the library's RCCL transport (grouped ncclSend / ncclRecv over xGMI): the run this
    pool's one-GPU boxes cannot make -- skipped unless the box has a GPU per rank.  Ready for the first multi-GPU node."""
    import torch
    if torch.cuda.device_count() < case[3]:
        pytest.skip(f"needs {case[3]} GPUs (RCCL refuses two ranks on one device); this box has {torch.cuda.device_count()}")
    _run_ranks(oracle, rank_exe, tmp_path, case, "rccl")


def _run_ranks(oracle, rank_exe, tmp_path, case, transport):
    from oracle.pyoracle import default_option
    w, h, d, world, batch, lead, steps = case
    seed = 0x7A110 + w
    env = dict(os.environ)
    if transport:
        env["SGM_TILES_TRANSPORT"] = transport
    procs = [subprocess.Popen([rank_exe, str(tmp_path), str(r), str(world), str(w), str(h), str(d), str(batch), str(lead), str(steps),
                               str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][0][-300:], outs[r][1][-1500:])
    opt = default_option(d)
    for k in range(steps):
        got = np.fromfile(os.path.join(str(tmp_path), f"step{k}.f32"), dtype=np.float32).reshape(batch, h, w)
        for j in range(batch):
            left, right = oracle.synth_pair(w, h, d, seed + k * batch + j)
            assert_same(got[j], oracle.run(left, right, opt)["final"], f"step {k} frame {j}")
