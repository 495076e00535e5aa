#!/usr/bin/env python
"""bench.py -- generate_fast() throughput of the MI355X engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3x64] [--samples 2000]

A "step" is one generate_fast()-shaped job: every stream of the workload generates ``--samples`` audio samples
(queue reset + ONE wn_generate job = one persistent kernel per chain, the chains running concurrently; temperature 1.0,
host-drawn uniforms, inputs resident in HBM).
``value`` = audio samples/s summed over all streams and all GPUs.  N > 1: one process per GPU (torchrun), streams
sharded across ranks with no data-path collective; the finished index blocks are gathered to rank 0 over RCCL inside
the timed region (that is the job's only exchange step).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {  # name -> (BASELINE.json config, streams per GPU)
    "cfg3x64": ("cfg3", 64),   # configs[2] / configs[3]: 64 independent streams per GPU
    "cfg3x1": ("cfg3", 1),     # the 16 kHz real-time target
    "cfg2x1": ("cfg2", 1),     # configs[1]
    "cfg1x1": ("cfg1", 1),     # configs[0]
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def time_workload(cfgname, n_streams, samples, steps, warmup, dist, device):
    from mi355_wavenet import engine, synth
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=0)
    eng = engine.Engine(cfg, W, n_streams=n_streams, device_index=device)
    rs = np.random.RandomState(1234 + (dist.get_rank() if dist else 0))
    first = eng.mem.upload(np.full((n_streams, 1), 128, dtype=np.int32))
    uni = eng.mem.upload(rs.random_sample((n_streams, samples)))
    out = eng.mem.empty((n_streams, samples), np.int32)
    gathered = None
    if dist and dist.get_rank() == 0:
        gathered = [torch.empty_like(out) for _ in range(dist.get_world_size())]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

    def one_step(i=None):
        eng.reset()
        if i is not None:
            ev0[i].record()
        eng.launch(first, 1, samples, 1.0, None, uni, out, None, timeout_ms=20000)
        if i is not None:
            ev1[i].record()
        if dist:
            dist.gather(out, gathered, dst=0)

    for _ in range(warmup):
        one_step()
    eng.wait()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(i)
    eng.wait()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t1 = time.perf_counter()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    info = eng.info()
    idx = out.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 256
    eng.close()
    return t1 - t0, kernel_ms, info, cfg


def train_step_cfg5(device, N=32, L=16000, reps=3):
    """BASELINE configs[4] (SURVEY.md 8d "cfg5"): one training step -- model(x), F.cross_entropy, backward, Adam -- at
    layers=10 blocks=5 128/128/512, N one-second 16 kHz clips given as class indices, through the facade's native
    matrix-core forward + backward.  Reports step time and executed TFLOP/s (forward GEMM work x 3)."""
    import wavenet_model
    torch.manual_seed(0)
    m = wavenet_model.WaveNetModel(layers=10, blocks=5, dilation_channels=128, residual_channels=128, skip_channels=512,
                                   end_channels=256, classes=256, output_length=1, kernel_size=2, bias=False).cuda(device)
    m.output_length = out_len = L - m.receptive_field + 1
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 256, (N, L), generator=g).cuda(device)
    target = torch.randint(0, 256, (N * out_len,), generator=g).cuda(device)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    R = D = 128; S = 512; E = 256; C = 256
    need, fwd = out_len, 0
    for d in reversed([2 ** (i % 10) for i in range(50)]):
        fwd += 2 * N * need * (2 * R * 2 * D + D * R) + 2 * N * out_len * D * S
        need += d
    fwd += 2 * N * out_len * (S * E + E * C)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m.train_forward_indices(idx), target)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    return {"ms_per_step": round(ms, 2), "clips": N, "clip_samples": L, "output_length": out_len, "dtype": "f32 (matrix cores)",
            "tflop_per_step": round(3 * fwd / 1e12, 2), "tflops": round(3 * fwd / ms / 1e9, 1),
            "mfma_f32_peak_frac": round(3 * fwd / ms / 1e9 / 157.3, 3), "loss": round(float(loss.detach()), 4)}


def train5_main(a, dist, rank, local, n_gpus, global_batch=32, L=16000):
    """--workload train5: BASELINE configs[4] -- the training step of layers=10 blocks=5 128/128/512 on one-second 16 kHz
    clips, GLOBAL batch 32 split over the ranks (plain data parallel): native forward + backward per rank, ONE
    flat gradient all-reduce over RCCL (wavenet_training.average_gradients), Adam on every rank.  A "step" is one
    optimiser step; value = clips (= seconds of audio) per second over all GPUs; scaling is strong (fixed global batch)."""
    import wavenet_model
    import wavenet_training
    torch.manual_seed(0)
    m = wavenet_model.WaveNetModel(layers=10, blocks=5, dilation_channels=128, residual_channels=128, skip_channels=512,
                                   end_channels=256, classes=256, output_length=1, kernel_size=2, bias=False).cuda(local)
    m.output_length = out_len = L - m.receptive_field + 1
    n_local = global_batch // n_gpus
    g = torch.Generator().manual_seed(1 + rank)
    idx = torch.randint(0, 256, (n_local, L), generator=g).cuda(local)
    target = torch.randint(0, 256, (n_local * out_len,), generator=g).cuda(local)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    group = dist.group.WORLD if dist else None

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m.train_forward_indices(idx), target)
        loss.backward()
        if dist:
            wavenet_training.average_gradients(m.parameters(), group)
        opt.step()

    for _ in range(max(a.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        R = D = 128; S = 512; E = 256; C = 256
        need, fwd = out_len, 0
        for d in reversed([2 ** (i % 10) for i in range(50)]):
            fwd += 2 * global_batch * need * (2 * R * 2 * D + D * R) + 2 * global_batch * out_len * D * S
            need += d
        fwd += 2 * global_batch * out_len * (S * E + E * C)
        ms = wall / a.steps * 1e3
        tflops = 3 * fwd / ms / 1e9
        print(json.dumps({
            "metric": "training step throughput, one-second 16 kHz clips per second (forward + backward + Adam), whole job",
            "value": round(global_batch / (ms * 1e-3), 2), "unit": "clips/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": max(a.warmup, 1),
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded random weights, uniform random class indices and targets)",
            "config": {"workload": "train5: WaveNetModel(layers=10, blocks=5, 128/128/512/256), global batch %d clips x %d samples, "
                                   "output_length %d, data parallel over %d GPU(s), one flat gradient all-reduce per step"
                                   % (global_batch, L, out_len, n_gpus), "global_batch": global_batch, "clips_per_gpu": n_local},
            "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": 157.3 * n_gpus, "unit": "TFLOP/s",
                         "frac": round(tflops / (157.3 * n_gpus), 4), "traffic": None,
                         "kernel": "wn_fwd_gemm / wn_bwd_gemm_tn (fp32 MFMA)", "flop_per_step": int(3 * fwd)}}))
    if dist:
        dist.destroy_process_group()


def cpu_baseline(cfgname, budget_s=10.0):
    """The reference's CPU path (torch restatement of generate_fast, oracle/restated.py, proven bit-equal to the real
    reference in tests/test_oracle_pinning.py) timed on this box's host cores: a bounded single-stream sample.  The path
    is framework-dispatch bound (203 tiny conv1d calls per sample), so it is timed with 1 thread and with torch's default
    thread count and the faster of the two is reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restated
    from mi355_wavenet import synth
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=0)
    r = restated.RestatedWaveNet(cfg, W)
    default_threads = torch.get_num_threads()
    best = None
    for threads in (1, default_threads):
        torch.set_num_threads(threads)
        np.random.seed(0)
        r.generate_fast(10, temperature=1.0, return_details=True)  # warm-up
        n = 40
        t0 = time.perf_counter()
        r.generate_fast(n, temperature=1.0, return_details=True)
        dt = time.perf_counter() - t0
        n2 = int(max(50, min(3000, budget_s / (dt / n))))
        t0 = time.perf_counter()
        r.generate_fast(n2, temperature=1.0, return_details=True)
        rate = n2 / (time.perf_counter() - t0)
        if best is None or rate > best[0]:
            best = (rate, threads, n2)
        if default_threads == 1:
            break
    torch.set_num_threads(default_threads)
    return {"value": round(best[0], 2), "unit": "samples/s", "cores": int(best[1]), "kind": "port",
            "sample": "%s single stream, %d samples of generate_fast(temperature=1.0) through oracle/restated.py "
                      "(op-for-op torch restatement of the reference's CPU path; best of 1 and %d torch threads, "
                      "host has %d logical cores)" % (cfgname, best[2], default_threads, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3x64", choices=sorted(WORKLOADS) + ["train5"],
                    help="train5: BASELINE configs[4], the data-parallel training step (global batch 32, strong scaling)")
    ap.add_argument("--samples", type=int, default=2000, help="audio samples per stream per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        dist = dist_mod
    else:
        torch.cuda.set_device(local)
    n_gpus = world if world > 1 else 1
    if a.gpus != n_gpus and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)

    if a.workload == "train5":
        train5_main(a, dist, rank, local, n_gpus)
        return
    cfgname, per_gpu = WORKLOADS[a.workload]
    wall, kernel_ms, info, cfg = time_workload(cfgname, per_gpu, a.samples, a.steps, a.warmup, dist, local)
    if dist:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    from mi355_wavenet import synth
    total_samples = n_gpus * per_gpu * a.samples * a.steps
    value = total_samples / wall
    bytes_per_tstep = synth.algorithmic_bytes_per_step(cfg, per_gpu)   # W_touched + streams*(Q+8)
    bytes_per_launch = bytes_per_tstep * a.samples
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path)).get(a.workload)
        if pmc:  # hand-off traffic is linear in the number of timesteps: scale to this launch
            traffic = int((pmc["fetch_kib"] + pmc["write_kib"]) * 1024 * a.samples / pmc["samples_per_launch"])
    line = {
        "metric": "generate_fast() audio samples/sec (256-class mu-law), whole job over all GPUs",
        "value": round(value, 1), "unit": "samples/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded random weights, host-drawn uniforms)",
        "config": {"workload": "%s: WaveNetModel(%s), %d independent streams per GPU x %d samples per step, "
                               "temperature 1.0" % (a.workload, ", ".join("%s=%s" % kv for kv in cfg.items()), per_gpu, a.samples),
                   "streams_per_gpu": per_gpu, "samples_per_stream_per_step": a.samples,
                   "per_stream_samples_per_s": round(value / (n_gpus * per_gpu), 1),
                   "chain": {k: info[k] for k in ("kernel_variant", "n_chains", "layer_split", "head_split", "n_workgroups", "lds_bytes")}},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "kernel": ("wn_generate_kernel_v2m" if per_gpu > 1 else "wn_generate_kernel_v2") if info["kernel_variant"] == 2 else "wn_generate_kernel", "kernel_ms_per_launch": round(kernel_ms, 3),
                     "algorithmic_bytes_per_launch": int(bytes_per_launch),
                     "algorithmic_bytes_per_timestep": int(bytes_per_tstep), "launches_per_step": 1,
                     "note": "one launch = one wn_generate job: n_chains persistent kernels running CONCURRENTLY (two per CU), timed "
                             "together with HIP events on the launch stream; rocprofv3 lists them as n_chains overlapping dispatches"},
    }
    if n_gpus == 1 and not a.no_extra:
        extra = {}
        for wl in ("cfg3x1", "cfg2x1"):
            if wl == a.workload:
                continue
            c2, s2 = WORKLOADS[wl]
            w2, k2, i2, cf2 = time_workload(c2, s2, 8000, 2, 1, None, local)
            extra[wl] = {"samples_per_s": round(2 * 8000 * s2 / w2, 1), "kernel_ms_per_launch": round(k2, 3),
                         "hbm_frac": round(synth.algorithmic_bytes_per_step(cf2, s2) * 8000 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "n_workgroups": i2["n_workgroups"]}
        line["extra"] = extra
    if n_gpus == 1 and not a.no_extra:
        try:
            line["extra"]["train_cfg5"] = train_step_cfg5(local)
        except Exception as e:  # noqa: BLE001 -- the secondary measurement must never cost the headline line
            line["extra"]["train_cfg5"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if n_gpus == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfgname)
    print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
