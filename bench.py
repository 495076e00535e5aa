#!/usr/bin/env python
"""bench.py -- generate_fast() throughput of the MI355X engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3x64] [--samples 16000]

A "step" is one ``WaveNetModel.generate_fast(--samples, first_samples=(streams, 1), temperature=1.0)`` call per GPU: every
stream of the workload generates ``--samples`` audio samples.  Two legs are timed with the same K / W:
  * facade (the headline ``value``): wall clock around the Python call -- uniform draw, H2D, queue reset, the persistent
    kernels, D2H, mu-law expansion;
  * engine_level: the same job through the C ABI with inputs and outputs resident in HBM (one wn_generate per step), with
    HIP events around the launch for the roofline figure.
``value`` = audio samples/s summed over all streams and all GPUs.  N > 1: one process per GPU (torchrun), streams
sharded across ranks with no data-path collective; the finished blocks are gathered to rank 0 over RCCL inside
the timed region (that is the job's only exchange step); per-rank kernel and gather times are reported.
``verified``: the last timed buffer of both legs re-checked against the C oracle (300 samples of two streams).
Prints ONE JSON line on rank 0.
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
    "cfg3x128": ("cfg3", 128), # the same model with twice the streams: the chain's throughput form at its best (not a BASELINE config)
    "cfg2x1": ("cfg2", 1),     # configs[1]
    "cfg1x1": ("cfg1", 1),     # configs[0]
    "cfg2x64": ("cfg2", 64),   # configs[1]'s model as a multi-stream job
    "chaconnex1": ("chaconne", 1),  # the only trained-model configuration in the reference tree (train_script.py:17-25: 32/32/1024/512, bias)
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
CPU_CELL_REPS = 3      # timed runs per cell of the CPU baseline matrix (the median is reported)


def _rank_stats(dist, kernel_ms, gather_ms, facade_ms=None):
    """Per-rank timings collected on rank 0 (so that a scaling run explains itself): [{rank, kernel_ms, gather_ms, ...}]."""
    mine = {"rank": dist.get_rank() if dist else 0, "kernel_ms": round(kernel_ms, 3), "gather_ms": round(gather_ms, 3)}
    if facade_ms is not None:
        mine["facade_ms"] = round(facade_ms, 3)
    if not dist:
        return [mine]
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    return everyone


BASELINE_METRIC = "generate_fast() audio samples/sec/GPU (256-class \u00b5-law) at 1/2/4/8 MI355X"   # BASELINE.json "metric", verbatim


def streams_by_rank(scaling, per_gpu, n_gpus):
    """Streams every rank runs: weak = the workload's streams on EVERY GPU; strong = BASELINE configs[3], 512 streams sharded over the ranks."""
    if scaling == "strong":
        from mi355_wavenet import streams
        return [hi - lo for lo, hi in (streams.shard_bounds(512, r, n_gpus) for r in range(n_gpus))]
    return [per_gpu] * n_gpus


def reduce_ranks(per_rank, rank_streams, samples, steps, fac_wall_max, eng_wall_max):
    """The figures of an N-rank run from the per-rank records (pure: tests/test_host_logic.py feeds it fake 8-rank stats).  per_rank: [{rank,
    kernel_ms, gather_ms, facade_ms}] (ms per STEP); rank_streams[r]: streams of rank r; *_wall_max: seconds for `steps` steps, MAX over the ranks
    (barrier + synchronize on both sides).  value = the WHOLE JOB's samples / the slowest rank's wall clock (the bench contract); value_per_gpu = that
    / N = the metric's own per-GPU figure; per rank: what each rank generated per second of ITS OWN wall clock (a straggler shows here)."""
    n = len(rank_streams)
    total_streams = int(sum(rank_streams))
    total_samples = total_streams * samples * steps
    out = {"total_streams": total_streams, "value": total_samples / fac_wall_max, "engine_value": total_samples / eng_wall_max}
    out["value_per_gpu"] = out["value"] / n
    ranks = []
    for rec in sorted(per_rank, key=lambda r: r["rank"]):
        r = dict(rec)
        st = rank_streams[rec["rank"]]
        r["streams"] = int(st)
        if rec.get("facade_ms"):
            r["samples_per_s"] = round(st * samples / (rec["facade_ms"] * 1e-3), 1)
        r["engine_samples_per_s"] = round(st * samples / ((rec["kernel_ms"] + rec["gather_ms"]) * 1e-3), 1) if rec["kernel_ms"] + rec["gather_ms"] > 0 else None
        ranks.append(r)
    out["per_rank"] = ranks
    return out


def time_engine(cfgname, n_streams, samples, steps, warmup, dist, device, temperature=1.0, weights=None):
    """Engine-level leg: one wn_generate job per step through the C ABI with every input already resident in HBM (first
    samples, host-drawn uniforms) and the indices left in HBM; N > 1: the finished index blocks are gathered to rank 0 over
    RCCL inside the timed region.  HIP events on the launch stream bracket the job (kernel_ms) and the gather (gather_ms)."""
    from mi355_wavenet import engine, synth
    cfg = synth.CONFIGS[cfgname]
    W = weights if weights is not None else synth.init_weights(cfg, seed=0)
    eng = engine.Engine(cfg, W, n_streams=n_streams, device_index=device)
    rank = dist.get_rank() if dist else 0
    rs = np.random.RandomState(1234 + rank)
    first_h = np.full((n_streams, 1), 128, dtype=np.int32)
    uni_h = rs.random_sample((n_streams, samples))
    first = eng.mem.upload(first_h)
    uni = eng.mem.upload(uni_h) if temperature > 0 else None   # temperature 0: the greedy branch (wavenet_model.py:290-294), no uniforms
    out = eng.mem.empty((n_streams, samples), np.int32)
    gathered = None
    if dist and rank == 0:
        gathered = [torch.empty_like(out) for _ in range(dist.get_world_size())]

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(steps)] for _ in range(3)]

    def one_step(i=None):
        eng.reset()
        if i is not None:
            ev[0][i].record()
        eng.launch(first, 1, samples, float(temperature), None, uni, out, None, timeout_ms=20000)
        if i is not None:
            ev[1][i].record()
        if dist:
            dist.gather(out, gathered, dst=0)
        if i is not None:
            ev[2][i].record()

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
    kernel_each = [a.elapsed_time(b) for a, b in zip(ev[0], ev[1])]
    kernel_ms = float(np.mean(kernel_each))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev[1], ev[2])]))
    info = eng.info()
    idx = out.cpu().numpy()
    eng.close()
    return {"wall": t1 - t0, "kernel_ms": kernel_ms, "kernel_ms_median": float(np.median(kernel_each)), "gather_ms": gather_ms, "info": info,
            "cfg": cfg, "W": W, "last_idx": idx, "first": first_h, "uniforms": uni_h}


def time_facade(cfgname, n_streams, samples, steps, warmup, dist, device):
    """Facade leg -- what the metric names: WaveNetModel.generate_fast(samples, first_samples=(streams, 1), temperature=1.0)
    per step, wall clock around the Python call: the uniform draw from the global numpy RNG, the H2D of first samples and
    uniforms, queue reset, the persistent-kernel jobs (the reference's contract cuts the job after generating step 100 for its
    timing print: two launches per call), the D2H of the indices and the mu-law expansion to float64 audio.
    N > 1: the finished audio of every rank is gathered to rank 0 over RCCL inside the timed region."""
    import contextlib
    import io
    import wavenet_model
    from mi355_wavenet import synth
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=0)
    m = wavenet_model.WaveNetModel(output_length=1, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    m = m.cuda(device)
    rank = dist.get_rank() if dist else 0
    first = torch.full((n_streams, 1), 128, dtype=torch.int64)
    sink = io.StringIO()
    gathered = None
    dev = torch.device("cuda", device)
    if dist and rank == 0:
        gathered = [torch.empty((n_streams, samples), dtype=torch.float64, device=dev) for _ in range(dist.get_world_size())]
    audio = None

    def one_step(seed=None):
        nonlocal audio
        if seed is not None:
            np.random.seed(seed)
        with contextlib.redirect_stdout(sink):
            audio = m.generate_fast(samples, first_samples=first, temperature=1.0)
        if dist:
            dist.gather(torch.from_numpy(audio).to(dev), gathered, dst=0)

    import gc
    for _ in range(max(warmup, 1)):
        one_step()
    # (harness hygiene, the same for this leg and for the CPU baseline: everything allocated so far -- torch, the model, the engine -- is moved
    #  out of the cyclic collector's reach (gc.freeze) so that a generation-2 pass over a process with torch loaded, tens of ms, does not land in a
    #  timed step; the collector itself STAYS ENABLED: what a generate_fast() caller pays per call is in the number.  Mean and median are reported.)
    gc.collect()
    gc.freeze()
    try:
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        marks = [t0]
        for i in range(steps):
            one_step(seed=4321 + 100 * rank + i)
            marks.append(time.perf_counter())  # (generate_fast returns host audio: every step ends synchronised)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t1 = time.perf_counter()
    finally:
        gc.unfreeze()
    step_ms = [1e3 * (b - a) for a, b in zip(marks, marks[1:])]
    # the uniforms generate_fast drew in the LAST step: same seed, same draw shapes ((streams, 100) then (streams, rest))
    np.random.seed(4321 + 100 * rank + steps - 1)
    cut = 100 if samples >= 100 else samples
    u = np.random.random_sample((n_streams, cut))
    if samples > cut:
        u = np.concatenate([u, np.random.random_sample((n_streams, samples - cut))], axis=1)
    return {"wall": t1 - t0, "step_ms_median": float(np.median(step_ms)), "cfg": cfg, "W": W, "last_audio": audio, "first": first.numpy(), "uniforms": u}


def verify_against_oracle(cfg, W, first, uniforms, idx=None, audio=None, streams=(0, -1), n=300, temperature=1.0):
    """Re-runs the C oracle (oracle/wn_oracle.c -- the checker, never the thing measured) for the first n samples of two
    streams of the buffer that was just timed: True iff the indices (or the expanded audio) are identical."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    ok = True
    ns = first.shape[0]
    for s in streams:
        s = s % ns
        n_s = min(n, uniforms.shape[1])
        o_idx, _ = c_oracle.generate(cfg, W, n_s, first[s], float(temperature), 0.0, uniforms[s, :n_s] if temperature > 0 else None, want_logits=False)
        if idx is not None:
            ok = ok and bool(np.array_equal(idx[s, :n_s], o_idx))
        if audio is not None:
            ok = ok and bool(np.array_equal(audio[s, :n_s], c_oracle.expand(o_idx)))
    return ok


def train_algorithmic_bytes(N, L, out_len, bf16, layers=10, blocks=5, R=128, D=128, S=512, E=256, C=256):
    """HBM bytes one training step has to move when every saved activation is written once and read once (DESIGN.md section 2c states the
    same formula).  Per layer l, on its M_l = N * need[l+1] rows (need = the trailing time steps the loss depends on):
      forward   read x_l (4R), write x_{l+1} (4R), write z (zb D), write the gate pair tanh | sigmoid (gb D)
      backward  read dx' (4R), read the gate pair (gb D), write and read [dF|dG] (2 fb 2D), read z (zb D: dWres), read x_l (4R: dWfg), write dx_l (4R)
    with zb / gb / fb bytes per element: 2 / 4 / 2 in the bf16 step (z, the packed pair and [dF|dG] stored as bf16), 4 / 8 / 4 in the fp32 step.
    Skip path on the Mo = N * output_length skip rows: z on the skip rows read twice where the layers left it (zb D per layer: the grouped skip product
    and its weight gradient; until round 5 a second copy was written for them), dskip read twice per block (sb S: the bf16 step reads its bf16 shadow,
    written once: 2S), dzg written and read (db D per layer: stored as bf16 in the bf16 step since round 5), skip read-modify-write per block (8S);
    head: logits and dlogits (4C each), e and de (2 x 4E each), skip and dskip once more (4S each).  Weights and their gradients (30 MB each) are
    noise next to the activations."""
    zb, gb, fb = (2, 4, 2) if bf16 else (4, 8, 4)
    sb, db = (2, 2) if bf16 else (4, 4)
    NL = layers * blocks
    dil = [2 ** (i % layers) for i in range(NL)]
    need = [0] * (NL + 1)
    need[NL] = out_len
    for l in range(NL - 1, -1, -1):
        need[l] = min(need[l + 1] + dil[l], L)
    total = 0
    for l in range(NL):
        M = N * need[l + 1]
        res = 1 if l < NL - 1 else 0
        total += M * (4 * R + 4 * R * res + zb * D + gb * D)                                     # forward
        total += M * (4 * R * res + gb * D + 2 * fb * 2 * D + zb * D * res + 4 * R + 4 * R)     # backward
    Mo = N * out_len
    total += Mo * NL * (2 * zb * D + 2 * db * D) + Mo * blocks * (2 * sb * S + 8 * S) + (Mo * 2 * S if bf16 else 0)
    total += Mo * (2 * 4 * C + 4 * 4 * E + 2 * 4 * S)
    return int(total)



# ---- the config-5 training fixture (tests/golden/golden_v6.npz: the REAL reference's loss, logit rows and gradient digests at N = 32, L = 16 000,
# output_length 10 885, on seeded inputs and informative weights -- tests/golden/make_golden.py --v6).  The training legs start from it and check one
# forward + backward against it before they time anything: `verified`.
CFG5_WSEED, CFG5_DSEED, CFG5_N, CFG5_L = 41, 42, 32, 16000


def cfg5_fixture():
    import zlib
    from mi355_wavenet import synth
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_v6.npz"))
    v6 = {k: z[k] for k in z.files}
    out_len = CFG5_L - synth.receptive_field(synth.CONFIGS["cfg3"]) + 1
    rs = np.random.RandomState(CFG5_DSEED)
    ids = rs.randint(0, 256, (CFG5_N, CFG5_L))
    target = rs.randint(0, 256, (CFG5_N, out_len))
    meta = [int(v) for v in v6["cfg5_meta"]]
    ok = meta[:5] == [CFG5_WSEED, CFG5_DSEED, CFG5_N, CFG5_L, out_len] and zlib.crc32(ids.astype(np.int16).tobytes()) == meta[5] and \
        zlib.crc32(target.astype(np.int16).tobytes()) == meta[6]
    if not ok:
        raise RuntimeError("tests/golden/golden_v6.npz does not describe the inputs bench.py regenerates")
    return v6, ids, target, out_len


def cfg5_model(device, out_len, precision):
    import wavenet_model
    from mi355_wavenet import synth
    cfg = synth.CONFIGS["cfg3"]
    m = wavenet_model.WaveNetModel(output_length=out_len, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.init_weights(cfg, seed=CFG5_WSEED).items()})
    m = m.cuda(device)
    m.matrix_precision = precision
    return m


def cfg5_verify(m, v6, idx, target, out_len, clip0, n_total, precision):
    """One forward + loss + backward of clips [clip0, clip0 + n) on the fixture's weights against the reference's record.  The whole batch on one GPU
    (clip0 = 0, n = n_total = 32): loss (1e-5 relative; the bf16 step: inside twice the spread of its oracle's evaluation orders), the stored logit rows
    (fp32: 1e-4; bf16: 1.5x / 2x the oracle's rms / max), every gradient's digest (fp32: 2e-5 of the tensor's largest element; bf16: 1.5x / 2x the
    oracle's rms / max deviation).  A shard of the batch (data parallel): its clips' logit rows; the gradients are checked by the caller AFTER the
    all-reduce.  Returns (ok, details, gradients-as-numpy or None)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest as dg
    from mi355_wavenet import training
    n = idx.shape[0]
    m.zero_grad(set_to_none=True)
    logits = m.train_forward_indices(idx)
    loss = training.cross_entropy(m._wn_train_runner, logits, target)
    loss.backward()
    torch.cuda.synchronize()
    rows = torch.as_tensor(v6["cfg5_logit_rows"], device=logits.device)
    got = logits.detach().view(n, out_len, 256)[:, rows, :].cpu().numpy().astype(np.float64)
    d = got - v6["cfg5_n32_logits"][clip0:clip0 + n]
    rms, mx = float(np.sqrt((d ** 2).mean())), float(np.abs(d).max())
    bf16 = precision != "fp32"
    noise = v6["cfg5_n32_bf16_noise"]
    det = {"logit_rows_rms": round(rms, 7), "logit_rows_max": round(mx, 7), "loss": round(float(loss.detach()), 6)}
    ok = (rms <= 1.5 * noise[:, 0].max() and mx <= 2.0 * noise[:, 1].max()) if bf16 else mx <= 1e-4
    if n != n_total:
        return ok, det, None
    ref_loss = float(v6["cfg5_n32_loss"][0])
    det["reference_loss"] = round(ref_loss, 6)
    dl = abs(float(loss.detach()) - ref_loss)
    ok = ok and (dl <= max(2.0 * noise[:, 2].max(), 2e-4) if bf16 else dl <= 1e-5 * ref_loss)
    g = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in m.named_parameters()}
    gok, gdet = cfg5_verify_grads(v6, g, precision)
    det.update(gdet)
    return ok and gok, det, g


def cfg5_verify_grads(v6, g, precision):
    """Gradient digests of the full-batch step against the fixture.  bf16 step: against the reference's fp32 digests, inside the spread of the bf16 oracle's
    evaluation orders.  fp32 step: against the reference's FLOAT64 evaluation, no further from it than the reference's own fp32 step is -- at this size
    a few hundred of the head's 267 M ReLU inputs lie within fp32 noise of zero, two correct fp32 evaluations disagree on those masks, and each flip moves
    the weight gradients by ~1e-3 of a tensor's largest element (tests/test_gpu_train_cfg5.py pins the gradients at 2e-5 on the rows whose masks are decided)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest as dg
    bf16 = precision != "fp32"
    head = "cfg5_n32_d_" if bf16 else "cfg5_n32_f64_d_"
    ref_d = {k[len(head):]: v for k, v in v6.items() if k.startswith(head)}
    got_d = dg.digest(g)
    devs = []
    for k, r in ref_d.items():
        if r[0] > 0:
            q = got_d[k]
            devs.append(max(abs(q[0] - r[0]) / r[0], abs(q[1] - r[1]) / r[1], float(np.abs(q[2:6] - r[2:6]).max()) / r[1], float(np.abs(q[6:] - r[6:]).max()) / r[0]))
    devs = np.array(devs)
    drms, dmax = float(np.sqrt((devs ** 2).mean())), float(devs.max())
    if bf16:
        noise = v6["cfg5_n32_bf16_noise"]
        ok = drms <= 1.5 * noise[:, 3].max() and dmax <= 2.0 * noise[:, 4].max()
        bar = "gradient digests vs the real reference's fp32 step: inside the spread of the bf16 oracle's evaluation orders at this size (rms x1.5, max x2: golden_v6 cfg5_n32_bf16_noise)"
    else:
        noise = v6["cfg5_n32_fp32_noise"]
        ok = drms <= 2.5 * noise[0] and dmax <= 2.5 * noise[1]
        bar = ("logit rows 1e-4, loss 1e-5 against the real reference; gradient digests vs the reference's float64 evaluation: no further than the reference's own fp32 "
               "step is (rms %.1e max %.1e), times 2.5 -- undecidable ReLU masks, see tests/test_gpu_train_cfg5.py" % (noise[0], noise[1]))
    return ok, {"gradient_digest_dev_rms": float("%.3g" % drms), "gradient_digest_dev_max": float("%.3g" % dmax), "bar": bar}


def train_step_cfg5(device, N=32, L=16000, reps=3, precision="fp32"):
    """BASELINE configs[4] (SURVEY.md 8d "cfg5"): one training step -- model(x), F.cross_entropy, backward, Adam -- at
    layers=10 blocks=5 128/128/512, N one-second 16 kHz clips given as class indices, through the facade's native
    matrix-core forward + backward.  Reports step time and executed TFLOP/s (forward GEMM work x 3)."""
    from mi355_wavenet.optim import FusedAdam
    assert (N, L) == (CFG5_N, CFG5_L)
    v6, ids, tgt, out_len = cfg5_fixture()
    m = cfg5_model(device, out_len, precision)   # the fixture's weights (synth gain-1 init: the step starts at loss 6.05, not at the ln 256 of near-zero logits)
    idx = torch.from_numpy(ids).to(torch.int32).cuda(device)
    target = torch.from_numpy(tgt.reshape(-1)).cuda(device)
    verified, vdet, _ = cfg5_verify(m, v6, idx, target, out_len, 0, N, precision)
    opt = FusedAdam(m.parameters(), lr=1e-4)   # torch.optim.Adam's step as the engine's optimiser kernels (mi355_wavenet/optim.py, pinned to torch's in tests/test_gpu_training.py)
    R = D = 128; S = 512; E = 256; C = 256
    need, fwd = out_len, 0
    for d in reversed([2 ** (i % 10) for i in range(50)]):
        fwd += 2 * N * need * (2 * R * 2 * D + D * R) + 2 * N * out_len * D * S
        need += d
    fwd += 2 * N * out_len * (S * E + E * C)
    from mi355_wavenet import training

    def step():  # WavenetTrainer.train_step: forward, the engine's fused loss (WavenetTrainer._loss), backward, optimizer
        opt.zero_grad(set_to_none=True)
        logits = m.train_forward_indices(idx)
        loss = training.cross_entropy(m._wn_train_runner, logits, target)
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
    stats = m.wn_stats()
    assert not stats["torch_fallbacks"] and stats["native_train_forward"] == reps + 3, "a timed training step left the native kernels: %r" % (stats,)
    peak = 157.3 if precision == "fp32" else 2500.0  # dense MFMA peaks, TFLOP/s (MI355X_MICROARCH.md)
    hbm = train_algorithmic_bytes(N, L, out_len, precision != "fp32")
    return {"ms_per_step": round(ms, 2), "clips": N, "clip_samples": L, "output_length": out_len,
            "verified": bool(verified), "verified_against": vdet, "loss_after_%d_steps" % (reps + 2): round(float(loss.detach()), 4),
            "hbm_algorithmic_bytes_per_step": hbm, "hbm_gbs": round(hbm / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "hbm_measured": _train_pmc(precision, N, L),
            "dtype": "f32 (matrix cores)" if precision == "fp32" else "bf16 matrix operands (forward, activation- and weight-gradient products), f32 accumulation, f32 residual stream; z, gates, [dF|dG] and the skip convs' share of dz stored as bf16",
            "tflop_per_step": round(3 * fwd / 1e12, 2), "tflops": round(3 * fwd / ms / 1e9, 1),
            "mfma_peak_tflops": peak, "mfma_peak_frac": round(3 * fwd / ms / 1e9 / peak, 4)}


def train5_main(a, dist, rank, local, n_gpus, global_batch=32, L=16000):
    """--workload train5: BASELINE configs[4] -- the training step of layers=10 blocks=5 128/128/512 on one-second 16 kHz
    clips, GLOBAL batch 32 split over the ranks (plain data parallel): native forward + backward per rank, ONE
    flat gradient all-reduce over RCCL (wavenet_training.average_gradients), Adam on every rank.  A "step" is one
    optimiser step; value = clips (= seconds of audio) per second over all GPUs; scaling is strong (fixed global batch)."""
    import wavenet_training
    from mi355_wavenet.optim import FusedAdam
    assert (global_batch, L) == (CFG5_N, CFG5_L) and global_batch % n_gpus == 0
    bf16 = getattr(a, "train_precision", "bf16") == "bf16"
    precision = "bf16" if bf16 else "fp32"
    v6, ids, tgt, out_len = cfg5_fixture()
    m = cfg5_model(local, out_len, precision)   # the fixture's weights on every rank
    n_local = global_batch // n_gpus
    clip0 = rank * n_local                      # this rank's shard of the fixture's batch
    idx = torch.from_numpy(ids[clip0:clip0 + n_local]).to(torch.int32).cuda(local)
    target = torch.from_numpy(tgt[clip0:clip0 + n_local].reshape(-1)).cuda(local)
    # `verified`: one forward + backward on the fixture against the real reference's record before anything is timed (cfg5_verify); data parallel: every
    # rank checks its clips' logit rows, and the gradients are checked AFTER the all-reduce -- the mean of the shards' gradients IS the batch's
    ok, vdet, g = cfg5_verify(m, v6, idx, target, out_len, clip0, global_batch, precision)
    if dist and n_gpus > 1:
        wavenet_training.average_gradients(m.parameters(), dist.group.WORLD)
        torch.cuda.synchronize()
        g = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in m.named_parameters()}
        gok, gdet = cfg5_verify_grads(v6, g, precision)
        vdet.update(gdet)
        flag = torch.tensor([1.0 if (ok and gok) else 0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item() > 0)
    opt = FusedAdam(m.parameters(), lr=1e-4)   # torch.optim.Adam's step as the engine's optimiser kernels (mi355_wavenet/optim.py, pinned to torch's in tests/test_gpu_training.py)
    group = dist.group.WORLD if dist else None

    from mi355_wavenet import training

    def step():
        opt.zero_grad(set_to_none=True)
        logits = m.train_forward_indices(idx)
        loss = training.cross_entropy(m._wn_train_runner, logits, target)  # WavenetTrainer._loss
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
    stats = m.wn_stats()
    assert not stats["torch_fallbacks"] and stats["native_train_forward"] == a.steps + max(a.warmup, 1) + 1, "a timed training step left the native kernels: %r" % (stats,)
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
        hbm = train_algorithmic_bytes(global_batch, L, out_len, bf16)   # (all ranks together: the activations shard with the batch)
        measured = _train_pmc("bf16" if bf16 else "fp32", global_batch, L) if n_gpus == 1 else None
        print(json.dumps({
            "metric": "training step throughput, one-second 16 kHz clips per second (forward + backward + Adam), whole job",
            "value": round(global_batch / (ms * 1e-3), 2), "value_per_gpu": round(global_batch / (ms * 1e-3) / n_gpus, 2), "unit": "clips/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": max(a.warmup, 1), "verified": bool(ok), "verified_against": vdet,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16 matrix operands, f32 accumulation and residual stream" if bf16 else "f32",
            "data": "synthetic (seeded gain-1 random weights, uniform random class indices and targets: the fixture of tests/golden/golden_v6.npz)",
            "config": {"workload": "train5: WaveNetModel(layers=10, blocks=5, 128/128/512/256), global batch %d clips x %d samples, "
                                   "output_length %d, data parallel over %d GPU(s), one flat gradient all-reduce per step"
                                   % (global_batch, L, out_len, n_gpus), "global_batch": global_batch, "clips_per_gpu": n_local},
            "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": (2500.0 if bf16 else 157.3) * n_gpus, "unit": "TFLOP/s",
                         "frac": round(tflops / ((2500.0 if bf16 else 157.3) * n_gpus), 4),
                         "traffic": None if measured is None else measured["bytes_per_step"], "traffic_kind": None if measured is None else measured["kind"],
                         "hbm": {"algorithmic_bytes_per_step": hbm, "achieved_gbs": round(hbm / (ms * 1e-3) / 1e9, 1), "peak_gbs": HBM_PEAK_GBS * n_gpus,
                                 "frac": round(hbm / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * n_gpus), 4),
                                 "note": "the step is a stream over saved activations: this is the roofline that binds (formula: bench.py train_algorithmic_bytes)"},
                         "kernel": "wn_fwd_layer_bf16 / wn_bwd_layer_bf16 / wn_bwd_wfg_bf16 / wn_bwd_gemm_tn_bf16 / wn_fwd_gemm_bf16 (bf16 MFMA; the products are HBM streams at K = 128-512: DESIGN.md section 6)" if bf16
                                   else "wn_fwd_gemm / wn_bwd_gemm_tn (fp32 MFMA)", "flop_per_step": int(3 * fwd)}}))
    if dist:
        dist.destroy_process_group()


def cpu_baseline(cfgname, budget_s=3.0):
    """The reference's CPU path (torch restatement of generate_fast, oracle/restated.py, proven bit-equal to the real reference in
    tests/test_oracle_pinning.py) timed on this box's host cores, as BASELINE.md section 3 plans it: a bounded single-stream sample of
    cfg1, cfg2 and cfg3, each with 1 torch thread and with torch's default thread count (~3 s of CPU work per cell).  The path is
    framework-dispatch bound (203 tiny conv1d calls per sample at cfg3): threads do not help.  Every cell is the MEDIAN of
    CPU_CELL_REPS timed runs; ``value`` is the headline configuration's best cell; ``matrix`` holds all of them."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restated
    from mi355_wavenet import synth
    import gc
    default_threads = torch.get_num_threads()
    matrix = {}
    best = None
    gc.collect()
    gc.freeze()   # (the same collector treatment as the timed facade steps: enabled, with what exists so far out of its reach)
    for name in ("cfg1", "cfg2", "cfg3"):
        cfg = synth.CONFIGS[name]
        r = restated.RestatedWaveNet(cfg, synth.init_weights(cfg, seed=0))
        for threads in sorted({1, default_threads}):
            torch.set_num_threads(threads)
            np.random.seed(0)
            r.generate_fast(5, temperature=1.0, return_details=True)  # warm-up
            n = 20
            t0 = time.perf_counter()
            r.generate_fast(n, temperature=1.0, return_details=True)
            dt = time.perf_counter() - t0
            n2 = int(max(20, min(3000, budget_s / CPU_CELL_REPS / (dt / n))))
            rates = []
            for _ in range(CPU_CELL_REPS):   # median of CPU_CELL_REPS timed runs: one run is at the mercy of the box's other tenants
                t0 = time.perf_counter()
                r.generate_fast(n2, temperature=1.0, return_details=True)
                rates.append(n2 / (time.perf_counter() - t0))
            rate = float(np.median(rates))
            matrix["%s/%d threads" % (name, threads)] = {"samples_per_s": round(rate, 2), "samples": n2, "runs": [round(v, 2) for v in rates]}
            if name == cfgname and (best is None or rate > best[0]):
                best = (rate, threads, n2)
    torch.set_num_threads(default_threads)
    gc.unfreeze()
    if best is None:  # the headline configuration is not one of the three (e.g. chaconne): report cfg3's
        k = max((k for k in matrix if k.startswith("cfg3")), key=lambda k: matrix[k]["samples_per_s"])
        best = (matrix[k]["samples_per_s"], int(k.split("/")[1].split()[0]), matrix[k]["samples"])
    return {"value": round(best[0], 2), "unit": "samples/s", "cores": int(best[1]), "kind": "port",
            "kind_rationale": "the reference tree (/root/reference, pure Python) does not exist on the GPU box; oracle/restated.py "
                              "replays its ATen op sequence and is pinned bit-equal to the real reference's generate_fast() "
                              "(tests/test_oracle_pinning.py, fixtures from tests/golden/make_golden.py)",
            "sample": "%s single stream, %d samples of generate_fast(temperature=1.0) through oracle/restated.py "
                      "(op-for-op torch restatement of the reference's CPU path; median of 3 runs per cell, best of 1 and %d torch threads, "
                      "host has %d logical cores)" % (cfgname, best[2], default_threads, os.cpu_count()),
            "matrix": matrix}


def _pmc_traffic(a, info, per_gpu):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE of one launch,
    separate passes), scaled to this launch's timesteps -- hand-off traffic is linear in them.  The file names the kernel, its form and
    the date it was measured; the figure is REFUSED (traffic null, the reason in traffic_source) when the library that just ran is not
    that kernel in that form: a stale constant must not pass for a measurement of this build."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if a.scaling != "weak" or not os.path.exists(path):
        return None, "no PMC passes for this workload / scaling"
    pmc = json.load(open(path)).get(a.workload)
    if not pmc:
        return None, "no PMC passes for workload %s in profiles/pmc_traffic.json" % a.workload
    want = pmc.get("form", {})
    have = {k: info.get(k) for k in ("kernel_variant", "streams_per_item", "head_replicas", "n_samplers", "n_workgroups")}
    if any(want.get(k) != have[k] for k in have) or pmc.get("streams") != per_gpu:
        return None, "REFUSED: profiles/pmc_traffic.json was measured on %r (%d streams), this run is %r (%d streams)" % (want, pmc.get("streams", -1), have, per_gpu)
    traffic = int((pmc.get("fetch_correction", 1.0) * pmc["fetch_kib"] + pmc["write_kib"]) * 1024 * a.samples / pmc["samples_per_launch"])
    return traffic, {"file": "profiles/pmc_traffic.json", "kernel": pmc.get("kernel"), "measured": pmc.get("date"), "summary": pmc.get("summary"),
                     "counters": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, %d timesteps per launch, scaled to %d; traffic = %.1f x FETCH_SIZE + WRITE_SIZE (%s)"
                                 % (pmc["samples_per_launch"], a.samples, pmc.get("fetch_correction", 1.0), pmc.get("calibration", "uncalibrated"))}


PMC_LIVE_TIMESTEPS = 2000   # timesteps of the job the live PMC passes count (hand-off traffic is linear in them; the figure is scaled to --samples)


def pmc_counter_from_db(db_path, counter):
    """(kernel name, mean counter value per dispatch in KiB, dispatches) of the ONE generation kernel in a rocprofv3 results database, or None
    (pure: tests/test_host_logic.py feeds it a hand-made table)."""
    import sqlite3
    rows = sqlite3.connect(db_path).execute(
        "select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like '%wn_generate_kernel%' group by kernel_name",
        (counter,)).fetchall()
    if len(rows) != 1 or not rows[0][1]:
        return None
    return rows[0]


def pmc_traffic_bytes(fetch_kib, write_kib, samples, counted_timesteps, fetch_correction=2.0):
    """HBM bytes of a launch of `samples` timesteps from the counters of a launch of `counted_timesteps`: fetch_correction x FETCH_SIZE + WRITE_SIZE (KiB),
    linear in the timesteps (profiles/r04_pmc_calibration.txt: on the hand-offs' access patterns FETCH_SIZE reports half of the sc1 misses)."""
    return int((fetch_correction * fetch_kib + write_kib) * 1024 * samples / counted_timesteps)


def _pmc_live(a, info, per_gpu, cfgname):
    """HBM bytes per launch MEASURED IN THIS RUN: two rocprofv3 passes -- `--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: the counters in their own passes,
    with --kernel-trace only, as MI355X_MICROARCH.md prescribes -- around `tools/rate.py <cfg> <streams> 2000 1` in child processes (rocprofv3 cannot
    wrap the process that is already running; the children build the same library's engine on this GPU and run the same one-kernel job, two launches each,
    behind the timed region).  traffic = 2 x FETCH_SIZE + WRITE_SIZE (profiles/r04_pmc_calibration.txt: on the hand-offs' access patterns WRITE_SIZE is exact
    and FETCH_SIZE reports half of the sc1 misses), KiB -> bytes, scaled to this launch's timesteps.  Returns (traffic, source) or (None, reason): the
    caller falls back to the committed passes, labelled as replayed."""
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("WN_BENCH_NO_LIVE_PMC") == "1" or a.scaling != "weak":
        return None, "live PMC passes switched off"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    t0 = time.perf_counter()
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            env = dict(os.environ, TMPDIR="/tmp", WN_TESTING="1")
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "rate.py"),
                       cfgname, str(per_gpu), str(PMC_LIVE_TIMESTEPS), "1"]
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
                dbs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
                if r.returncode != 0 or not dbs:
                    return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stdout.decode(errors="replace")[-300:])
                row = pmc_counter_from_db(dbs[0], counter)
                if row is None:
                    return None, "rocprofv3 --pmc %s: no single generation kernel in the counter table" % counter
                got[counter] = row
    except Exception as e:   # (a profiler problem must not cost the bench line: the caller replays the committed passes instead)
        return None, "live PMC passes failed: %r" % (e,)
    fetch, write = got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]
    traffic = pmc_traffic_bytes(fetch, write, a.samples, PMC_LIVE_TIMESTEPS)
    return traffic, {"kernel": got["FETCH_SIZE"][0], "fetch_kib_per_launch": round(fetch, 1), "write_kib_per_launch": round(write, 1),
                     "dispatches_averaged": [got["FETCH_SIZE"][2], got["WRITE_SIZE"][2]], "timesteps_per_counted_launch": PMC_LIVE_TIMESTEPS,
                     "seconds": round(time.perf_counter() - t0, 1),
                     "counters": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes in child processes behind the timed region, %d timesteps per "
                                 "launch, scaled to %d; traffic = 2.0 x FETCH_SIZE + WRITE_SIZE (profiles/r04_pmc_calibration.txt: WRITE_SIZE exact, FETCH_SIZE reports half "
                                 "of the missed bytes for 8- and 16-byte sc1 loads)" % (PMC_LIVE_TIMESTEPS, a.samples)}


def _train_pmc(precision, N, L):
    """FETCH_SIZE + WRITE_SIZE of one training step summed over its kernels, from the committed rocprofv3 PMC passes of tools/collect_train_profiles.sh
    (profiles/pmc_traffic_train.json) -- replayed, like roofline.traffic of the generation line; None when there are none for this precision / size."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_train.json")
    if not os.path.exists(path):
        return None
    doc = json.load(open(path)).get("train5_%s" % precision)
    if not doc or doc.get("clips") != N or doc.get("clip_samples") != L:
        return None
    return {"bytes_per_step": int((doc.get("fetch_correction", 1.0) * doc["fetch_kib"] + doc["write_kib"]) * 1024), "kind": "replayed from profiles/pmc_traffic_train.json (%s, %s)" % (doc.get("date"), doc.get("summary"))}


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: re-executes this command line as N ranks, one per GPU, under
    ``python -m torch.distributed.run`` (rendezvous on 127.0.0.1, a free port) and returns its exit code.  Refuses (exit 2)
    when the node has fewer than N GPUs -- N ranks on fewer devices would be reported as an N-GPU number."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print("bench.py: --gpus %d but this node exposes %d GPU(s)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3x64", choices=sorted(WORKLOADS) + ["train5"],
                    help="train5: BASELINE configs[4], the data-parallel training step (global batch 32, strong scaling)")
    ap.add_argument("--samples", type=int, default=16000, help="audio samples per stream per step (SURVEY.md 8d: 16 000 = one second of audio)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's streams PER GPU (64 for cfg3x64); strong: BASELINE configs[3], 512 streams in total "
                         "sharded over the GPUs (512 / N per GPU)")
    ap.add_argument("--train-precision", default="bf16", choices=["bf16", "fp32"],
                    help="train5 only: matrix operand precision (BASELINE configs[4] names bf16 MFMA; fp32 = the parity default of the facade)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run through the distributed code path (RCCL process group, pick_device, gather, all-reduce of the timings) even "
                         "with ONE rank: the multi-GPU path exercised on a 1-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC passes (replayed) instead of two rocprofv3 passes behind the timed region")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_launch_ranks(a.gpus))  # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "WORLD_SIZE" not in os.environ:  # --force-dist without a launcher: a one-rank group of our own on 127.0.0.1
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        torch.cuda.set_device(local)
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        dist = dist_mod
        from mi355_wavenet import streams
        streams.pick_device(dist, local, torch.cuda.device_count())  # refuses two ranks on one GPU
    else:
        torch.cuda.set_device(local)
    n_gpus = world if world > 1 else 1
    if a.gpus != n_gpus:  # a launcher that started another number of ranks than --gpus says: never print a line that lies about N
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: refusing to report a %d-GPU figure as %d GPUs" % (a.gpus, world, n_gpus, a.gpus),
                  file=sys.stderr)
        sys.exit(2)

    if a.workload == "train5":
        train5_main(a, dist, rank, local, n_gpus)
        return
    cfgname, per_gpu = WORKLOADS[a.workload]
    if a.scaling == "strong":
        from mi355_wavenet import streams
        lo, hi = streams.shard_bounds(512, rank, n_gpus)
        per_gpu = hi - lo
    eng_leg = time_engine(cfgname, per_gpu, a.samples, a.steps, a.warmup, dist, local)
    fac_leg = time_facade(cfgname, per_gpu, a.samples, a.steps, a.warmup, dist, local)
    verified = verify_against_oracle(eng_leg["cfg"], eng_leg["W"], eng_leg["first"], eng_leg["uniforms"], idx=eng_leg["last_idx"]) and \
        verify_against_oracle(fac_leg["cfg"], fac_leg["W"], fac_leg["first"], fac_leg["uniforms"], audio=fac_leg["last_audio"])
    walls = [eng_leg["wall"], fac_leg["wall"], 0.0 if verified else 1.0]
    if dist:
        t = torch.tensor(walls, device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        walls = [float(v) for v in t.tolist()]
    per_rank = _rank_stats(dist, eng_leg["kernel_ms"], eng_leg["gather_ms"], fac_leg["wall"] / a.steps * 1e3)
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    eng_wall, fac_wall, any_bad = walls
    kernel_ms, info, cfg = eng_leg["kernel_ms"], eng_leg["info"], eng_leg["cfg"]

    from mi355_wavenet import synth
    red = reduce_ranks(per_rank, streams_by_rank(a.scaling, WORKLOADS[a.workload][1], n_gpus), a.samples, a.steps, fac_wall, eng_wall)
    total_streams, engine_value, per_rank = red["total_streams"], red["engine_value"], red["per_rank"]
    value, wall = red["value"], fac_wall  # the metric names generate_fast(): the facade's figure is the headline
    bytes_per_tstep = synth.algorithmic_bytes_per_step(cfg, per_gpu)   # W_touched + streams*(Q+8)
    bytes_per_launch = bytes_per_tstep * a.samples
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_source, traffic_kind = None, None, None
    if n_gpus == 1 and not dist and not a.no_live_pmc:
        traffic, traffic_source = _pmc_live(a, info, per_gpu, WORKLOADS[a.workload][0])
        traffic_kind = "measured in this run: rocprofv3 PMC passes of the same job (same library, same GPU) in child processes, scaled to this launch's timesteps"
    if traffic is None:
        live_reason = traffic_source
        traffic, traffic_source = _pmc_traffic(a, info, per_gpu)
        traffic_kind = None if traffic is None else ("replayed: PMC passes of an EARLIER run of this kernel in this form, scaled to this launch's timesteps -- not a "
                                                     "measurement of this run (live passes: %s)" % (live_reason or "not attempted"))
    kname = {1: "wn_generate_kernel", 3: "wn_generate_kernel_v3m", 4: "wn_generate_kernel_v4"}.get(info["kernel_variant"], "?")
    line = {
        "metric": BASELINE_METRIC,
        "value": round(value, 1), "value_per_gpu": round(red["value_per_gpu"], 1),
        "value_is": "the WHOLE JOB: samples of all streams on all %d GPU(s) per second of the slowest rank's wall clock (the bench contract); value_per_gpu = value / n_gpus "
                    "is the metric's per-GPU figure; per_rank[].samples_per_s is each rank against its own clock" % n_gpus,
        "unit": "samples/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 3), "median_ms_per_step": round(fac_leg["step_ms_median"], 3),
        "higher_is_better": True, "scaling": a.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded random weights, uniforms from the global numpy RNG)",
        "verified": bool(verified and any_bad == 0.0),
        "config": {"workload": "%s: WaveNetModel(%s).generate_fast(%d, first_samples=(%d, 1), temperature=1.0) per GPU per step"
                               % (a.workload, ", ".join("%s=%s" % kv for kv in cfg.items()), a.samples, per_gpu),
                   "streams_per_gpu": per_gpu, "samples_per_stream_per_step": a.samples,
                   "per_stream_samples_per_s": round(value / total_streams, 1),
                   "timed": "wall clock around the facade call: RNG draw, H2D, queue reset, kernels, D2H, mu-law expansion; the cyclic collector stays "
                            "enabled (objects that exist before the timed steps are gc.freeze()-d, here and in the CPU baseline); ms_per_step is the mean, median_ms_per_step the median"
                            + ("; + RCCL gather of the audio to rank 0" if dist else ""),
                   "chain": {k: info[k] for k in ("kernel_variant", "n_chains", "layer_split", "head_split", "n_workgroups", "lds_bytes",
                                                  "streams_per_item", "head_replicas", "n_samplers", "dev_overrides")}},
        "engine_level": {"value": round(engine_value, 1), "ms_per_step": round(eng_wall / a.steps * 1e3, 3),
                         "note": "the same job through the C ABI with inputs and outputs resident in HBM (one wn_generate per step"
                                 + ("; + RCCL gather of the index blocks" if dist else "") + ")",
                         "facade_over_engine": round(fac_wall / eng_wall, 4)},
        "per_rank": per_rank,
        "rccl_ranks": dist.get_world_size() if dist else 0, "dist_backend": dist.get_backend() if dist else None,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "traffic_kind": traffic_kind,
                     "traffic_source": traffic_source,
                     "traffic_over_algorithmic": None if traffic is None else round(traffic / bytes_per_launch, 3),
                     "kernel": kname, "kernel_ms_per_launch": round(kernel_ms, 3), "kernel_ms_per_launch_median": round(eng_leg["kernel_ms_median"], 3),
                     "algorithmic_bytes_per_launch": int(bytes_per_launch),
                     "algorithmic_bytes_per_timestep": int(bytes_per_tstep), "launches_per_step": 1,
                     "note": "engine-level leg: one launch = one wn_generate job of %d timesteps = ONE persistent kernel (n_chains > 1: the "
                             "rounds of a job beyond one chain's capacity, one after the other), timed with HIP events on the launch stream" % a.samples},
    }
    if n_gpus == 1 and not a.no_extra:
        extra = {}
        def extra_leg(c2, s2, n2, temperature=1.0, weights=None, check=True):
            leg = time_engine(c2, s2, n2, 2, 1, None, local, temperature=temperature, weights=weights)
            rec = {"samples_per_s": round(2 * n2 * s2 / leg["wall"], 1), "samples_per_stream": n2, "kernel_ms_per_launch": round(leg["kernel_ms"], 3),
                   "hbm_frac": round(synth.algorithmic_bytes_per_step(leg["cfg"], s2) * n2 / (leg["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                   "n_workgroups": leg["info"]["n_workgroups"], "kernel_variant": leg["info"]["kernel_variant"]}
            if check:
                rec["verified"] = verify_against_oracle(leg["cfg"], leg["W"], leg["first"], leg["uniforms"], idx=leg["last_idx"], streams=(0,), temperature=temperature)
            return rec

        for wl in ("cfg3x1", "cfg2x1", "cfg1x1", "chaconnex1", "cfg2x64", "cfg3x128"):
            if wl == a.workload:
                continue
            c2, s2 = WORKLOADS[wl]
            extra[wl] = extra_leg(c2, s2, 16000 if s2 == 1 else 2000)   # single streams: 16 000 samples = one second of audio
        # the greedy branch (temperature = 0: argmax, no uniforms) and PyTorch's DEFAULT init (logits dominated by end_conv_2.bias;
        # the arithmetic per timestep is the same -- reported for comparability with the survey's CPU probe) on the headline workload
        c2, s2 = WORKLOADS[a.workload]
        extra[a.workload + "_greedy"] = extra_leg(c2, s2, a.samples, temperature=0.0)
        import wavenet_model
        torch.manual_seed(0)
        default_w = {k: v.detach().numpy() for k, v in wavenet_model.WaveNetModel(**synth.CONFIGS[c2]).state_dict().items()}
        extra[a.workload + "_default_init"] = extra_leg(c2, s2, a.samples, weights=default_w)
        line["extra"] = extra
        try:
            line["extra"]["prime_generate_script_shape"] = prime_timing(local)
        except Exception as e:  # noqa: BLE001
            line["extra"]["prime_generate_script_shape"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        for key, prec in (("train_cfg5", "fp32"), ("train_cfg5_bf16", "bf16")):
            try:
                line["extra"][key] = train_step_cfg5(local, precision=prec)
            except Exception as e:  # noqa: BLE001 -- the secondary measurement must never cost the headline line
                line["extra"][key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if n_gpus == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfgname)
    print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


def prime_timing(device):
    """generate_script.py's call shape (/root/reference/generate_script.py:19-33): 5116 given samples at cfg3.  Time of the
    batched priming pass (wn_prime) next to the same priming through the per-sample chain."""
    from mi355_wavenet import engine, synth
    cfg = synth.CONFIGS["cfg3"]
    W = synth.init_weights(cfg, seed=0)
    eng = engine.Engine(cfg, W, n_streams=1, device_index=device)
    first = np.random.RandomState(7).randint(0, 256, (1, 5116))
    out = {}
    for name, batched in (("batched_ms", True), ("chain_ms", False)):
        eng.generate(1, first, temperature=0.0, batched_prime=batched)  # warm (workspace allocation)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(1, first, temperature=0.0, batched_prime=batched)
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) * 1e3, 3)
    eng.close()
    out["given_samples"] = 5116
    return out


if __name__ == "__main__":
    main()
