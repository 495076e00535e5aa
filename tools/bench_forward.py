"""Times the matrix-core forward (wn_forward) at BASELINE config 5's shape: layers=10 blocks=5 dil/res=128 skip=512,
N=32 one-second 16 kHz clips, output_length = 16000 - 5116 + 1 = 10885, next to the facade's torch path on the same GPU.

    python tools/bench_forward.py [N] [L]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import wavenet_model  # noqa: E402
from mi355_wavenet import _abi, engine, synth  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if len(args) > 0 else 32
    L = int(args[1]) if len(args) > 1 else 16000
    cfg = synth.CONFIGS["cfg3"]
    rf = synth.receptive_field(cfg)
    out_len = L - rf + 1
    W = synth.init_weights(cfg, seed=0)
    lib = _abi.Library(os.environ["WN_DEV_LIB"]) if os.environ.get("WN_DEV_LIB") else None  # dev build of the library
    eng = engine.Engine(cfg, W, lib=lib)
    ids = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (N, L))).cuda().int()
    R, D, S, E, C = 128, 128, 512, 256, 256
    dil = synth.dilation_list(cfg)
    need = out_len
    flops = 0
    for d in reversed(dil):
        flops += 2 * N * need * (2 * R * 2 * D + D * R) + 2 * N * out_len * D * S
        need += d
    flops += 2 * N * out_len * (S * E + E * C)
    for _ in range(2):
        y = eng.forward_indices(ids, out_len)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    ev0.record()
    for _ in range(reps):
        y = eng.forward_indices(ids, out_len)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    print("wn_forward N=%d L=%d out_len=%d: %.2f ms, %.2f TFLOP executed -> %.1f TFLOP/s fp32 MFMA (peak 157.3; frac %.3f)" % (
        N, L, out_len, ms, flops / 1e12, flops / ms / 1e9, flops / ms / 1e9 / 157.3))
    if "--fp32-only" in sys.argv:
        return
    try:
        eng.set_forward_precision(True)
        for _ in range(2):
            yb = eng.forward_indices(ids, out_len)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            yb = eng.forward_indices(ids, out_len)
        ev1.record()
        torch.cuda.synchronize()
        msb = ev0.elapsed_time(ev1) / reps
        print("wn_forward bf16 operands: %.2f ms -> %.1f TFLOP/s (bf16 dense peak 2500; frac %.3f); max |dlogit| vs fp32 kernel %.3g (scale %.3g)" % (
            msb, flops / msb / 1e9, flops / msb / 1e9 / 2500, float((yb - y).abs().max()), float(y.abs().max())))
        eng.set_forward_precision(False)
    except Exception as e:  # noqa: BLE001
        print("bf16 path unavailable:", e)
    if "--no-torch" in sys.argv:
        return
    # torch path of the facade on the same GPU (the reference's algorithm with ATen/MIOpen ops)
    m = wavenet_model.WaveNetModel(output_length=out_len, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    m = m.cuda()
    n_t = min(N, 8)
    x = torch.zeros(n_t, 256, L, device="cuda").scatter_(1, ids[:n_t].long().view(n_t, 1, L), 1.)
    with torch.no_grad():
        for _ in range(2):
            yt = m.wavenet(x, dilation_func=m.wavenet_dilate)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            yt = m.wavenet(x, dilation_func=m.wavenet_dilate)
        torch.cuda.synchronize()
        t_torch = (time.perf_counter() - t0) / 2 * 1e3
    ref = yt[:, :, -out_len:].transpose(1, 2).reshape(n_t * out_len, 256)
    dev = float((y[:n_t * out_len] - ref).abs().max())
    print("torch path (N=%d): %.2f ms -> scaled to N=%d: %.1f ms; max |dlogit| native vs torch-GPU %.3g" % (n_t, t_torch, N, t_torch * N / n_t, dev))


if __name__ == "__main__":
    main()
