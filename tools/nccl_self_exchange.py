#!/usr/bin/env python3
"""Smoke test of the RCCL code path of DeviceSlotEngine.exchange on a ONE-GPU box: a world-size-1 nccl group in which the
rank sends to itself and receives from itself in one grouped batch_isend_irecv (the only RCCL point-to-point a single GPU
can do).  Exercises exactly the calls the multi-GPU pipeline makes -- P2POp list, one group, work.wait() under the
communication stream, event ordering against a slot's own stream -- not the xGMI transport."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import soc_project_stereo_matching_amd as S  # noqa: E402
from soc_project_stereo_matching_amd.tiling import DeviceSlotEngine  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w, h, d = 320, 96, 64
eng = DeviceSlotEngine(0, w, h, S.default_option(d), (0, h), 3, host_staged=False)
src = eng.boundary(0, True, False)
dst = eng.boundary(1, True, True)
for it in range(5):
    with torch.cuda.stream(eng.stream[0]):
        src.fill_(it + 1)                               # produced on slot 0's stream
    eng.exchange(dist, [("send", src, 0), ("recv", dst, 0)], [0, 1])
    with torch.cuda.stream(eng.stream[1]):
        got = dst.clone()                               # consumed on slot 1's stream, ordered behind the exchange by an event
    eng.stream[1].synchronize()
    assert int(got.min()) == it + 1 and int(got.max()) == it + 1, (it, int(got.min()), int(got.max()))
eng.close()
dist.destroy_process_group()
print("nccl self-exchange ok: grouped isend/irecv through the communication stream, 5 rounds, event-ordered")
