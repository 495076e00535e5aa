"""-m gpu, needs >= 2 GPUs (skipped otherwise): the N > 1 path on the real backend -- two ranks, backend "nccl" (= RCCL over
xGMI), streams sharded by mi355_wavenet.streams, the finished index blocks gathered to rank 0 -- against the oracle; and
bench.py launched the way the driver launches it (python -m torch.distributed.run --nproc-per-node 2).  On a 1-GPU box only
the device-collision guard is exercised (two ranks on one device must be refused, not silently serialised) -- and, since round 4,
the SAME code paths in a process group of ONE rank on the real backend (`init_process_group("nccl")`, `pick_device`, the RCCL
gather of the index blocks, the flat gradient all-reduce, `bench.py --force-dist`): the driver's first 8-GPU run must not be the
first time this code meets RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu

two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "pytorch-wavenet_amd"), os.path.join(ROOT, "oracle"), HERE, env.get("PYTHONPATH", "")])
    return env


WORKER = r"""
import os, sys, json
import numpy as np, torch
import torch.distributed as dist
local = int(os.environ["LOCAL_RANK"])
forced = os.environ.get("WN_TEST_FORCE_DEVICE")
torch.cuda.set_device(local if forced is None else int(forced))
backend = os.environ.get("WN_TEST_BACKEND", "nccl")
if backend == "nccl":
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
else:
    dist.init_process_group(backend="gloo")
from mi355_wavenet import streams, synth
cfg = synth.CONFIGS["cfg2"]
W = synth.init_weights(cfg, seed=91)
rs = np.random.RandomState(91)
S, N = 7, 120   # odd: ranks get 4 and 3 streams
first = rs.randint(0, 256, (S, 9))
u = rs.random_sample((S, N))
try:
    out = streams.generate_streams(cfg, W, first, N, temperature=1.0, uniforms=u, dist=dist,
                                   device_index=None if forced is None else int(forced))
except RuntimeError as e:
    print("REFUSED:", e)
    dist.destroy_process_group()
    sys.exit(0)
if dist.get_rank() == 0:
    np.save(os.environ["WN_TEST_OUT"], out)
dist.barrier()
dist.destroy_process_group()
"""


def _run(tmp_path, extra_env, nproc=2):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = _env()
    env.update(extra_env)
    env["WN_TEST_OUT"] = str(tmp_path / "out.npy")
    port = 29800 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600), env["WN_TEST_OUT"]


def test_one_rank_over_rccl_matches_the_oracle(tmp_path):
    """The N-GPU job in a group of ONE rank over RCCL on the real device: init_process_group("nccl"), streams.pick_device (LOCAL_RANK),
    the shard of all 7 streams, the gather of the index block through dist.gather on a CUDA tensor -- result == oracle."""
    import c_oracle
    from mi355_wavenet import synth
    res, out_path = _run(tmp_path, {}, nproc=1)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "REFUSED" not in res.stdout
    out = np.load(out_path)
    cfg = synth.CONFIGS["cfg2"]
    W = synth.init_weights(cfg, seed=91)
    rs = np.random.RandomState(91)
    first = rs.randint(0, 256, (7, 9))
    u = rs.random_sample((7, 120))
    for s in range(7):
        idx, _ = c_oracle.generate(cfg, W, 120, first[s], 1.0, 0.0, u[s])
        assert np.array_equal(out[s], idx), s


GRAD_WORKER = r"""
import os, sys
import numpy as np, torch
import torch.distributed as dist
local = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
assert dist.get_backend() == "nccl"
import wavenet_model, wavenet_training
torch.manual_seed(5)
m = wavenet_model.WaveNetModel(layers=3, blocks=2, dilation_channels=32, residual_channels=32, skip_channels=64, end_channels=64,
                               classes=256, output_length=8).cuda(local)
g = torch.Generator().manual_seed(6 + dist.get_rank())
idx = torch.randint(0, 256, (2, m.receptive_field + 7), generator=g).cuda(local)
target = torch.randint(0, 256, (2 * 8,), generator=g).cuda(local)
loss = torch.nn.functional.cross_entropy(m.train_forward_indices(idx), target)   # native forward + backward
loss.backward()
before = [p.grad.clone() for p in m.parameters() if p.grad is not None]
wavenet_training.average_gradients(m.parameters(), dist.group.WORLD, always=True)   # ONE flat all-reduce over RCCL
after = [p.grad for p in m.parameters() if p.grad is not None]
world = dist.get_world_size()
if world == 1:
    assert all(torch.equal(a, b) for a, b in zip(before, after)), "mean over one rank must be the identity"
else:  # every rank holds the same mean afterwards
    flat = torch.cat([a.reshape(-1) for a in after])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref)
t = torch.tensor([float(len(after))], device="cuda")
dist.all_reduce(t)
print("GRADS_OK", world, int(t.item()))
dist.barrier()
dist.destroy_process_group()
"""


def test_gradient_all_reduce_over_rccl_in_a_group_of_one(tmp_path):
    """wavenet_training.average_gradients on the real backend (always=True: the all-reduce runs although one rank has nothing to average),
    behind the native training step."""
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER)
    port = 29700 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "GRADS_OK 1" in res.stdout


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_bench_one_gpu_through_the_distributed_path(launcher):
    """bench.py --gpus 1 --force-dist: the N-GPU code path (RCCL group, pick_device, gather of index blocks and audio inside the timed
    region, all-reduce of the timings) with one rank; the line says which backend carried it."""
    env = {k: v for k, v in _env().items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--samples", "400", "--no-extra", "--no-cpu-baseline"]
    if launcher == "plain":
        cmd = [sys.executable] + tail
    else:
        port = 29600 + os.getpid() % 1000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["dist_backend"] == "nccl"
    assert line["verified"] is True and line["value"] > 0
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["gather_ms"] >= 0
    assert "RCCL gather" in line["config"]["timed"]


@two_gpus
def test_two_ranks_over_rccl_match_the_oracle(tmp_path):
    import c_oracle
    from mi355_wavenet import synth
    res, out_path = _run(tmp_path, {})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = np.load(out_path)
    cfg = synth.CONFIGS["cfg2"]
    W = synth.init_weights(cfg, seed=91)
    rs = np.random.RandomState(91)
    first = rs.randint(0, 256, (7, 9))
    u = rs.random_sample((7, 120))
    for s in range(7):
        idx, _ = c_oracle.generate(cfg, W, 120, first[s], 1.0, 0.0, u[s])
        assert np.array_equal(out[s], idx), s


def test_two_ranks_on_one_device_are_refused(tmp_path):
    """streams.pick_device: both ranks forced onto device 0 (gloo rendezvous so that it also runs on a 1-GPU box)."""
    res, _ = _run(tmp_path, {"WN_TEST_FORCE_DEVICE": "0", "WN_TEST_BACKEND": "gloo"})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("REFUSED:") == 2 and "share HIP device 0" in res.stdout


@two_gpus
def test_bench_two_gpus_as_the_driver_launches_it():
    port = 29900 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--samples", "400"]
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert len(line["per_rank"]) == 2 and all(r["kernel_ms"] > 0 for r in line["per_rank"])
    assert line["verified"] is True


@two_gpus
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver may call it): bench.py re-executes itself as two ranks."""
    env = {k: v for k, v in _env().items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--samples", "400"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(line["per_rank"]) == 2 and line["verified"] is True


def test_bench_refuses_more_gpus_than_the_node_has():
    """... and never reports an N-GPU figure it did not measure: --gpus beyond the node's devices exits non-zero without a JSON line."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in _env().items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--samples", "100"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert "exposes" in res.stderr
    # a launcher that started ANOTHER number of ranks than --gpus says is refused as well (one rank, --gpus 2)
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--samples", "100"],
                         env=env1, capture_output=True, text=True, timeout=600)
    assert res.returncode == 2 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
