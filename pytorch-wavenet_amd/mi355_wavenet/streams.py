"""Many independent generation streams, sharded over the GPUs of a node.

Each stream is an independent autoregressive chain with private dilation queues, so the job partitions by stream with
NO collective on the data path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), every rank
runs its block of streams on its own engine, and the only exchange is the gather of the finished class-index blocks to
rank 0.  A single stream cannot be sharded (52 dependent hops per sample): replicas only.
"""
import numpy as np


def shard_bounds(n_total, rank, world):
    """Contiguous block of streams for `rank`: sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def generate_streams(cfg, weights, first_samples, num_samples, temperature=1.0, regularize=0.0, uniforms=None,
                     dist=None, device_index=None, lib=None):
    """first_samples (S, n_given) ints, uniforms (S, num_samples) float64 or None (greedy).
    Returns int32 (S, num_samples) on rank 0 (and on every rank when dist is None), else None."""
    from . import engine
    first_samples = np.asarray(first_samples)
    S = first_samples.shape[0]
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    lo, hi = shard_bounds(S, rank, world)
    mine = None
    if hi > lo:
        if device_index is None:
            device_index = 0
        eng = engine.Engine(cfg, weights, n_streams=hi - lo, device_index=device_index, lib=lib)
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
