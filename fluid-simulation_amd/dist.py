"""Host-side control plane of multi-GPU runs (one process per GPU, torch.distributed).

Only small host messages travel here: the RCCL unique id, timings, gathered planes for checks.  The
data path (halo planes per half-sweep, residual all-reduce) is RCCL inside the library.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_control_plane(backend="gloo"):
    """Joins the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def broadcast_bytes(payload, src=0):
    """Ships a bytes object from `src` to every rank (the RCCL unique id)."""
    if not dist.is_initialized():
        return payload
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def max_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def exchange_plane(mine, peer):
    """Swaps one plane (numpy array) with `peer`; lower rank sends first."""
    theirs = torch.empty_like(torch.from_numpy(mine))
    buf = torch.from_numpy(np.ascontiguousarray(mine))
    if dist.get_rank() < peer:
        dist.send(buf, peer); dist.recv(theirs, peer)
    else:
        dist.recv(theirs, peer); dist.send(buf, peer)
    return theirs.numpy()


def gather_planes(owned, dst=0):
    """Concatenates every rank's owned planes (along axis 0) on rank `dst`."""
    if not dist.is_initialized():
        return owned
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(owned, parts, dst=dst)
    return np.concatenate(parts, axis=0) if dist.get_rank() == dst else None
