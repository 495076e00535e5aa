"""Many independent generation streams, sharded over the GPUs of a node.

Each stream is an independent autoregressive chain with private dilation queues, so the job partitions by stream with
NO collective on the data path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), every rank
runs its block of streams on its own engine, and the only exchange is the gather of the finished class-index blocks to
rank 0.  A single stream cannot be sharded (52 dependent hops per sample): replicas only.
"""
import os
import socket

import numpy as np


def shard_bounds(n_total, rank, world):
    """Contiguous block of streams for `rank`: sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pick_device(dist=None, device_index=None, n_devices=None):
    """The HIP device of this rank: ``device_index`` when given, else LOCAL_RANK (torchrun), else rank modulo the visible
    devices.  With a process group the choice is checked across ranks: two ranks of one host on the same device would
    silently serialise on it (and share its 256 CUs between two persistent chains that each assume the whole chip)."""
    if device_index is None:
        if "LOCAL_RANK" in os.environ:
            device_index = int(os.environ["LOCAL_RANK"])
        elif dist is not None:
            device_index = dist.get_rank() % max(1, n_devices or 1)
        else:
            device_index = 0
    if n_devices is not None and not (0 <= device_index < n_devices):
        raise RuntimeError("rank wants HIP device %d but only %d are visible" % (device_index, n_devices))
    if dist is not None and dist.get_world_size() > 1:
        mine = (socket.gethostname(), int(device_index))
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        clash = sorted(r for r, v in enumerate(everyone) if everyone.count(v) > 1)
        if clash:
            raise RuntimeError("ranks %s share HIP device %d on %s: launch one process per GPU (torchrun sets LOCAL_RANK) or pass "
                               "device_index" % (clash, everyone[clash[0]][1], everyone[clash[0]][0]))
    return device_index


def generate_streams(cfg, weights, first_samples, num_samples, temperature=1.0, regularize=0.0, uniforms=None,
                     dist=None, device_index=None, lib=None, mem=None):
    """first_samples (S, n_given) ints, uniforms (S, num_samples) float64 or None (greedy).
    Returns int32 (S, num_samples) on rank 0 (and on every rank when dist is None), else None.
    device_index: this rank's HIP device (default: LOCAL_RANK; checked for collisions across ranks, see pick_device)."""
    from . import engine
    first_samples = np.asarray(first_samples)
    S = first_samples.shape[0]
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    lo, hi = shard_bounds(S, rank, world)
    mine = None
    if mem is None:
        import torch
        device_index = pick_device(dist, device_index, torch.cuda.device_count())
    elif device_index is None:
        device_index = 0  # an injected memory provider (tests): no device to pick
    if hi > lo:
        eng = engine.Engine(cfg, weights, n_streams=hi - lo, device_index=device_index, lib=lib, mem=mem)
        u = None if uniforms is None else np.asarray(uniforms)[lo:hi]
        mine = eng.generate(num_samples, first_samples[lo:hi], temperature=temperature, regularize=regularize, uniforms=u)
        eng.close()
    if dist is None:
        return mine
    import torch
    width = -(-S // world)  # every rank contributes a block of the same (padded) height
    block = np.zeros((width, num_samples), dtype=np.int32)
    if mine is not None:
        block[:hi - lo] = mine
    backend = dist.get_backend()
    t = torch.from_numpy(block)
    if backend == "nccl":
        t = t.cuda(device_index)
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)
    if rank != 0:
        return None
    out = np.zeros((S, num_samples), dtype=np.int32)
    for r in range(world):
        a, b = shard_bounds(S, r, world)
        out[a:b] = gathered[r].cpu().numpy()[:b - a]
    return out
