"""Frame-level sharding of a stereo stream over the GPUs of a node (one process per GPU).

Frames are independent units of the hot path, so the data path needs no collective:
rank r owns every frame whose index is congruent to r modulo the world size, matches them on its own
GPU, and -- only if the caller wants the results in one place -- the disparity maps are gathered to
rank 0 with one torch.distributed tensor gather (device tensors over RCCL, host tensors over gloo).
`matcher` is any callable (left, right) -> float32 disparity: in the product it is an `SGMStream` over libsgm_mi355x.so; the CPU tests plug in the oracle to exercise
this host logic with the gloo backend.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np


def frames_of_rank(n_frames: int, world: int, rank: int) -> List[int]:
    """Round-robin assignment: frame i -> rank i % world (keeps a live stream balanced)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, n_frames, world))


def owner_of_frame(index: int, world: int) -> int:
    return index % world


class SGMStream:
    """Round-robins frames of one rank over a few `SGMInstance`s (HIP streams) of its GPU so that
    several frames are in flight; every frame is SGM_Reset + SGM_Match."""

    def __init__(self, device: int, width: int, height: int, option, in_flight: int = 4):
        from .sgm import SGMInstance
        self.w, self.h, self.option = width, height, option
        self.instances = [SGMInstance(device) for _ in range(max(1, in_flight))]
        for inst in self.instances:
            if not inst.reset(width, height, option):
                raise RuntimeError("sgm_reset failed")
        self._next = 0

    def submit_device(self, d_left: int, d_right: int, d_out: int):
        """Asynchronous: device pointers in, result lands in d_out after synchronize()."""
        inst = self.instances[self._next % len(self.instances)]
        self._next += 1
        if not inst.reset(self.w, self.h, self.option):
            raise RuntimeError("sgm_reset failed")
        if not inst.match_device(d_left, d_right, d_out):
            raise RuntimeError("sgm_match_device failed")
        return inst

    def __call__(self, left: np.ndarray, right: np.ndarray) -> np.ndarray:
        inst = self.instances[self._next % len(self.instances)]
        self._next += 1
        if not inst.reset(self.w, self.h, self.option):
            raise RuntimeError("sgm_reset failed")
        out = inst.match(left, right)
        if out is None:
            raise RuntimeError("sgm_match failed")
        return out

    def synchronize(self):
        for inst in self.instances:
            if not inst.synchronize():
                raise RuntimeError("sgm_synchronize failed")

    def close(self):
        for inst in self.instances:
            inst.close()


def match_sharded(frames: Sequence[Tuple[np.ndarray, np.ndarray]], matcher: Callable, world: int, rank: int,
                  gather_to_rank0: bool = True, dist=None) -> Optional[List[np.ndarray]]:
    """Match the frames this rank owns; optionally gather all disparity maps (in frame order) on rank 0.

    `dist` is torch.distributed (already initialised) when world > 1.  Returns the full list on rank 0
    (or the rank's own results when gather_to_rank0 is False), None on the other ranks."""
    mine = frames_of_rank(len(frames), world, rank)
    local = {}
    for i in mine:
        m = matcher(frames[i][0], frames[i][1])
        if m is None:                                             # a failed match must not travel as a map of zeros / NaNs
            raise RuntimeError(f"matcher returned no result for frame {i} (rank {rank})")
        local[i] = m
    if not gather_to_rank0:
        return [local[i] for i in mine]
    if world == 1:
        return [local[i] for i in range(len(frames))]
    # one tensor gather of the ranks' maps (padded to the largest share); nothing is pickled through the host.  The BACKEND decides
    # where the gathered tensors live, the same on every rank whatever its matcher returned and however many frames it owns:
    # RCCL ("nccl") moves device tensors only, so host maps (SGMStream returns numpy) are uploaded to the rank's GPU first; gloo
    # moves host tensors, so device maps are downloaded.
    import torch
    most = -(-len(frames) // world)
    first = next(iter(local.values())) if local else None
    shape = [most] + list(first.shape if first is not None else np.shape(frames[0][0]))
    on_device = dist.get_backend() == "nccl"
    if on_device:
        dev = first.device if (torch.is_tensor(first) and first.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    mine_t = torch.zeros(shape, dtype=torch.float32, device=dev)
    for k, i in enumerate(mine):
        m = local[i] if torch.is_tensor(local[i]) else torch.from_numpy(np.ascontiguousarray(local[i], dtype=np.float32))
        if tuple(m.shape) != tuple(shape[1:]):
            raise RuntimeError(f"frame {i}: the matcher returned a map of shape {tuple(m.shape)}, expected {tuple(shape[1:])}")
        mine_t[k] = m.to(dev)
    # how many frames each rank really matched travels with the maps: rank 0 checks that no frame is missing
    count = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.empty_like(count) for _ in range(world)] if rank == 0 else None
    dist.gather(count, counts, dst=0)
    parts = [torch.empty_like(mine_t) for _ in range(world)] if rank == 0 else None
    dist.gather(mine_t, parts, dst=0)
    if rank != 0:
        return None
    for r in range(world):
        want = len(frames_of_rank(len(frames), world, r))
        if int(counts[r].item()) != want:
            raise RuntimeError(f"rank {r} matched {int(counts[r].item())} of its {want} frames")
    out = []
    for i in range(len(frames)):
        m = parts[i % world][i // world]                     # frame i is the (i // world)-th frame of rank i % world
        out.append(m if on_device else m.numpy())
    return out
