"""Multi-GPU plumbing: frames are independent, so they shard across ranks with NO data-path collective.

The only exchange is at set-up: rank 0 packs the shared branch table + decoder config into one small blob
(vit_hip_pack_blob) and broadcasts it -- over RCCL/xGMI when the process group is "nccl" (which IS RCCL on ROCm), over
gloo in the CPU tests.  Every rank then builds an identical decoder from the blob (vit_hip_create_from_blob), mirroring the
reference's "one branch table shared by many decoders" (README.md:14, examples/run_benchmark.cpp:193-197).
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(total_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [begin, end) of the frames owned by `rank`; sizes differ by at most one frame."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(total_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def broadcast_blob(blob: Optional[bytes], src: int = 0, device=None) -> bytes:
    """broadcast the decoder blob from `src` to every rank of the default process group; returns it on all ranks."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        assert blob is not None
        return blob
    rank = dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def gather_frames(local, dst: int = 0):
    """gather per-rank result tensors [f_local][...] on `dst` (list in rank order there, None elsewhere): result
    collection only, never on the timed path."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size()
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    out = None
    if dist.get_rank() == dst:
        out = [torch.empty((int(c.item()),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) for c in counts]
    # point-to-point keeps ragged shards simple
    if dist.get_rank() == dst:
        out[dst].copy_(local)
        for r in range(world):
            if r != dst and out[r].numel():
                dist.recv(out[r], src=r)
    elif local.numel():
        dist.send(local.contiguous(), dst=dst)
    return out
