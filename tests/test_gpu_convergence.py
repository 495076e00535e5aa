"""Does the bf16 step TRAIN?  (/root/reference/wavenet_training.py:58-90 is a loop, and until round 6 the bf16 step was
only ever compared one step at a time.)

250 optimiser steps of WavenetTrainer.train_step -- FusedAdam, gradient clipping -- on a learnable synthetic mu-law signal (three sines and a little
noise through audio_data.quantize_data) with a 20-layer model at config 5's widths (128 / 128 / 512 / 256), three times from the same initial weights
on the same batches: the facade's torch path (the reference's conv1d graph under torch autograd, WN_TORCH_BACKWARD=1), the native fp32 step, the
native bf16 step.  Asserted:
  * native fp32 tracks the torch path's loss curve while two fp32 trajectories CAN track each other: 5e-4 relative over the first 10 steps, 5e-3 over the
    first 30.  Not further: Adam's first steps are sign-like (g / sqrt(v) with v ~ g^2), so two fp32 evaluations of the SAME step that differ in the
    last bits of a near-zero gradient element take visibly different steps there, and neither path is bit-reproducible (fp32 atomics here, MIOpen's
    backward there).  Measured on MI355X, two runs (profiles/r06_convergence_curves.json): bit-equal losses for 4 steps, 1.3e-4 by step 10, 6.6e-4 /
    1.6e-3 by step 30, 2e-3 / 8e-3 by step 50, then per-step differences of up to 4-9 % on a loss that itself moves +-5 % from batch to batch;
  * so the rest is asserted on means: over the last 50 steps native fp32 is within 3 % of the torch path and bf16 within 3 % of fp32 (measured: 0.3-1.5 %
    and 0.3-0.7 %), no step of either is further than 15 % from its counterpart, and all three end below 60 % of their initial loss
    (measured: 34 %: 5.59 -> 1.89).
The curves go to gpurun_out/r06_convergence_curves.json when that directory exists (committed as profiles/r06_convergence_curves.json).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))

pytestmark = pytest.mark.gpu

CFG = dict(layers=10, blocks=2, dilation_channels=128, residual_channels=128, skip_channels=512, end_channels=256, classes=256, kernel_size=2, bias=False)
STEPS, BATCH, OUT_LEN = 250, 8, 1024


def _signal(n, seed=0):
    from audio_data import quantize_data
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    x = 0.45 * np.sin(2 * np.pi * 220.0 * t) + 0.25 * np.sin(2 * np.pi * 331.7 * t + 0.3) + 0.15 * np.sin(2 * np.pi * 523.3 * t + 1.1)
    x = x * (0.75 + 0.25 * np.sin(2 * np.pi * 1.7 * t)) + 0.01 * rs.standard_normal(n)
    return quantize_data(np.clip(x, -1, 1), 256).astype(np.int64)


def _run(mode, stream, starts):
    import wavenet_model
    import wavenet_training
    from mi355_wavenet import synth
    from mi355_wavenet.optim import FusedAdam
    m = wavenet_model.WaveNetModel(output_length=OUT_LEN, **CFG)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.init_weights(CFG, seed=77).items()})
    m = m.cuda()
    m.matrix_precision = "bf16" if mode == "bf16" else "fp32"
    L = m.receptive_field + OUT_LEN - 1
    trainer = wavenet_training.WavenetTrainer(m, dataset=None, optimizer=FusedAdam, lr=1e-3, gradient_clipping=1.0)
    dev_stream = torch.from_numpy(stream).cuda()
    ar = torch.arange(L + 1, device="cuda")
    losses = []
    if mode == "torch":
        os.environ["WN_TORCH_BACKWARD"] = "1"
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)   # (the torch path announces itself once: wanted here)
            for s in range(STEPS):
                win = dev_stream[torch.as_tensor(starts[s], device="cuda").unsqueeze(1) + ar.unsqueeze(0)]   # (BATCH, L + 1)
                idx, target = win[:, :L], win[:, -OUT_LEN:].reshape(-1)
                if mode == "torch":
                    x = torch.zeros(BATCH, 256, L, device="cuda").scatter_(1, idx.unsqueeze(1), 1.0)
                    losses.append(trainer.train_step("onehot", x, target))
                else:
                    losses.append(trainer.train_step("indices", idx.to(torch.int32), target))
    finally:
        os.environ.pop("WN_TORCH_BACKWARD", None)
    st = m.wn_stats()
    if mode == "torch":
        assert st["native_train_forward"] == 0 and sum(st["torch_fallbacks"].values()) == STEPS
    else:
        assert st["native_train_forward"] == STEPS and not st["torch_fallbacks"]
    return np.array(losses)


def test_the_bf16_step_trains_like_the_fp32_step_and_the_torch_path():
    import wavenet_model
    m0 = wavenet_model.WaveNetModel(output_length=OUT_LEN, **CFG)
    L = m0.receptive_field + OUT_LEN - 1
    stream = _signal(400000)
    rs = np.random.RandomState(5)
    starts = rs.randint(0, len(stream) - L - 1, (STEPS, BATCH))
    curves = {mode: _run(mode, stream, starts) for mode in ("torch", "fp32", "bf16")}
    t, f, b = curves["torch"], curves["fp32"], curves["bf16"]
    tail = slice(STEPS - 50, STEPS)
    rel = np.abs(f - t) / t
    summary = {"steps": STEPS, "batch": BATCH, "clip_samples": int(L), "output_length": OUT_LEN, "model": CFG,
               "initial_loss": {k: float(v[0]) for k, v in curves.items()}, "final_loss_mean_of_last_50": {k: float(v[tail].mean()) for k, v in curves.items()},
               "fp32_vs_torch_relative": {"first_10_max": float(rel[:10].max()), "first_30_max": float(rel[:30].max()), "all_max": float(rel.max()), "last_50_mean": float(abs(f[tail].mean() - t[tail].mean()) / t[tail].mean())},
               "bf16_vs_fp32_relative": {"last_50_mean": float(abs(b[tail].mean() - f[tail].mean()) / f[tail].mean()), "all_max": float((np.abs(b - f) / f).max())},
               "curves": {k: [round(float(x), 6) for x in v] for k, v in curves.items()}}
    print(json.dumps({k: v for k, v in summary.items() if k != "curves"}))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r06_convergence_curves.json"), "w") as fh:
            json.dump(summary, fh)
    assert rel[:10].max() <= 5e-4 and rel[:30].max() <= 5e-3, (rel[:10].max(), rel[:30].max())
    assert rel.max() <= 0.15 and float((np.abs(b - f) / f).max()) <= 0.15
    assert summary["fp32_vs_torch_relative"]["last_50_mean"] <= 3e-2
    assert summary["bf16_vs_fp32_relative"]["last_50_mean"] <= 3e-2
    for k, v in curves.items():
        assert v[tail].mean() < 0.6 * v[0], (k, float(v[0]), float(v[tail].mean()))
