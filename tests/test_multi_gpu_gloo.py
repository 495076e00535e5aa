"""N > 1 path on CPU: world_size 2 over gloo.  Stream sharding + gather of mi355_wavenet.streams, with the host-memory
test double of the C ABI standing in for the two GPUs (explicitly injected; test infrastructure only)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "pytorch-wavenet_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from double_lib import double_backend, double_library
    from mi355_wavenet import streams, synth
    cfg = synth.CONFIGS["tiny_bias"]
    W = synth.init_weights(cfg, seed=81)
    rs = np.random.RandomState(81)
    S, N = 5, 40  # odd stream count: ranks get 3 and 2
    first = rs.randint(0, 256, (S, 7))
    u = rs.random_sample((S, N))
    out = streams.generate_streams(cfg, W, first, N, temperature=1.0, uniforms=u, dist=dist, **double_backend())
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_streams_shard_and_gather_world2():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from mi355_wavenet import streams, synth
    assert [streams.shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [streams.shard_bounds(512, r, 8) for r in range(8)][-1] == (448, 512)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    cfg = synth.CONFIGS["tiny_bias"]
    W = synth.init_weights(cfg, seed=81)
    rs = np.random.RandomState(81)
    first = rs.randint(0, 256, (5, 7))
    u = rs.random_sample((5, 40))
    for s in range(5):
        idx, _ = c_oracle.generate(cfg, W, 40, first[s], 1.0, 0.0, u[s])
        assert np.array_equal(out[s], idx), s


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N without WORLD_SIZE re-executes itself under torch.distributed.run with N ranks on 127.0.0.1 (host logic
    only: the launch itself is exercised on a 2-GPU box in tests/test_gpu_multi.py)."""
    import importlib
    import subprocess
    import sys
    import torch
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench._launch_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench._launch_ranks(4) == 2   # fewer devices than ranks: refused
